"""Pins (SGG_ABI_VERSION, digest of every prototype of include/sgg_hip.h) in tests/abi.lock.

A prototype change without a version bump lets a stale libsgg_hip.so pass `_lib.load()`'s version check and be called
with shifted arguments.  `python tools/abi_lock.py` records the current pair; it refuses to re-point an already recorded
version at a new digest (bump SGG_ABI_VERSION in the header and ABI_VERSION in sgg_amd/_lib.py first).
tests/test_abi_cpu.py::test_abi_version_is_bumped_with_the_prototypes checks the pair."""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOCK = os.path.join(ROOT, 'tests', 'abi.lock')


def prototypes():
    """{name: canonical argument-type string} of every `int sgg_*(...)` prototype of the header."""
    txt = open(os.path.join(ROOT, 'include', 'sgg_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    out = {}
    for name, args in re.findall(r'\b(sgg_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;', txt, flags=re.S):
        types = []
        for a in ([] if args.strip() in ('', 'void') else args.split(',')):
            a = ' '.join(a.split())
            a = re.sub(r'\s*\b[A-Za-z_][A-Za-z0-9_]*$', '', a) if not a.endswith('*') else a   # drop the parameter name
            types.append(a.replace(' *', '*'))
        out[name] = ','.join(types)
    return out


def header_version():
    return int(re.search(r'#define\s+SGG_ABI_VERSION\s+(\d+)', open(os.path.join(ROOT, 'include', 'sgg_hip.h')).read()).group(1))


def digest():
    p = prototypes()
    return hashlib.sha256('\n'.join('%s(%s)' % (k, p[k]) for k in sorted(p)).encode()).hexdigest()[:16]


def main():
    lock = json.load(open(LOCK)) if os.path.exists(LOCK) else {}
    v, d = str(header_version()), digest()
    if lock.get(v, d) != d:
        sys.exit('version %s is already locked to %s, the header now hashes to %s: bump SGG_ABI_VERSION' % (v, lock[v], d))
    lock[v] = d
    json.dump(lock, open(LOCK, 'w'), indent=1, sort_keys=True)
    print('abi.lock: version %s -> %s' % (v, d))


if __name__ == '__main__':
    main()
