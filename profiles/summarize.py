#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel stats table (text).
usage: python profiles/summarize.py <results.db> [out.txt]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by %s order by 3 desc" % (namecol, namecol)).fetchall()
    total = sum(r[2] for r in rows)
    lines = ['%-112s %7s %12s %10s %10s %10s %6s' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', '%')]
    for n, c, t, a, mn, mx in rows:
        lines.append('%-112s %7d %12.1f %10.2f %10.2f %10.2f %6.2f' % (short(n), c, t / 1e3, a / 1e3, mn / 1e3, mx / 1e3,
                                                                      100.0 * t / total))
    out = '\n'.join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(out + '\n')
    print(out)


if __name__ == '__main__':
    main()
