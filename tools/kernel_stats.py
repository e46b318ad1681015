"""rocprofv3 --kernel-trace results (.db, rocpd schema) -> the per-kernel summary kept under profiles/.
    python tools/kernel_stats.py gpurun_out/<dir>/<name>_results.db > profiles/rNN_<name>_kernel_stats.txt"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
rows = db.execute('select %s, start, end, grid_x, workgroup_x from kernels' % name_col).fetchall()
agg = collections.defaultdict(list)
shapes = collections.defaultdict(list)
for name, s, e, gx, wx in rows:
    short = name.replace('void ', '').replace('(anonymous namespace)::', '')
    agg[short].append((e - s) / 1e3)
    if 'mfma_' in short or 'imp_' in short or 'roi_align' in short:
        shapes[(short[:60], gx // max(wx, 1))].append((e - s) / 1e3)
total = sum(sum(v) for v in agg.values())
print('%-110s %9s %12s %10s %10s %10s %6s' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', '%'))
for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print('%-110s %9d %12.1f %10.2f %10.2f %10.2f %6.2f' % (name[:110], len(v), sum(v), sum(v) / len(v), min(v), max(v),
                                                           100 * sum(v) / total))

# the same kernel symbol serves several contractions: split the big ones by launch grid (workgroups), so that e.g. the fc6 weight
# gradient (1568 workgroups of the ping-pong kernel) and fc6 on the edges (496) can be read off separately
print()
print('%-60s %10s %9s %12s %10s %10s %10s' % ('kernel, by grid', 'workgroups', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us'))
for (name, wgs), v in sorted(shapes.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) / len(v) < 50:
        continue
    print('%-60s %10d %9d %12.1f %10.2f %10.2f %10.2f' % (name, wgs, len(v), sum(v), sum(v) / len(v), min(v), max(v)))

# several contractions share a grid too (fc6 on the edges, K = 25600, and fc7, K = 4096, are both 496 workgroups of the ping-pong
# kernel): the launches of at least 1 ms of such a row are listed on their own -- bench.py's `roofline` kernel is one of these rows
print()
print('%-60s %10s %9s %12s %10s %10s %10s' % ('launches >= 1000 us of a mixed row', 'workgroups', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us'))
for (name, wgs), v in sorted(shapes.items(), key=lambda kv: -sum(kv[1])):
    big = [x for x in v if x >= 1000.0]
    if big and len(big) < len(v):
        print('%-60s %10d %9d %12.1f %10.2f %10.2f %10.2f' % (name, wgs, len(big), sum(big), sum(big) / len(big), min(big), max(big)))
