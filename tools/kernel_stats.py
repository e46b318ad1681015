"""rocprofv3 --kernel-trace results (.db, rocpd schema) -> the per-kernel summary kept under profiles/.
    python tools/kernel_stats.py gpurun_out/<dir>/<name>_results.db > profiles/rNN_<name>_kernel_stats.txt"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
rows = db.execute('select %s, start, end from kernels' % name_col).fetchall()
agg = collections.defaultdict(list)
for name, s, e in rows:
    agg[name.replace('void ', '').replace('(anonymous namespace)::', '')].append((e - s) / 1e3)
total = sum(sum(v) for v in agg.values())
print('%-110s %9s %12s %10s %10s %10s %6s' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', '%'))
for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print('%-110s %9d %12.1f %10.2f %10.2f %10.2f %6.2f' % (name[:110], len(v), sum(v), sum(v) / len(v), min(v), max(v),
                                                           100 * sum(v) / total))
