"""rocprofv3 --kernel-trace .db -> how much of the steady-state timeline has NO kernel running, and the longest idle gaps with the kernels
around them (host-bound or dependency-bound stretches).   python tools/exp/timeline_gaps.py <results.db> [t0_frac t1_frac]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
rows = db.execute('select %s, start, end from kernels order by start' % name_col).fetchall()
# window: from the k0-th to the k1-th launch of the per-step image prep kernel (steady-state steps)
k0, k1 = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (20, 30)
marks = [r[1] for r in rows if 'image_prep_batch_kernel' in r[0]]
a, b = marks[k0], marks[k1]
sel = [r for r in rows if r[1] >= a and r[1] < b]
print('%d steps' % (k1 - k0))
busy, gaps, cur_end, last = 0, [], None, None
for name, s, e in sel:
    if cur_end is None:
        cur_s, cur_end, last = s, e, name
        continue
    if s > cur_end:
        busy += cur_end - cur_s
        gaps.append((s - cur_end, last, name))
        cur_s, cur_end = s, e
    else:
        cur_end = max(cur_end, e)
    if e >= cur_end:
        last = name
busy += cur_end - cur_s
span = sel[-1][2] - sel[0][1]
print('window %.1f ms: %d kernels, busy %.1f ms (%.1f %%), idle %.2f ms in %d gaps' % (span / 1e6, len(sel), busy / 1e6, 100.0 * busy / span,
                                                                                   (span - busy) / 1e6, len(gaps)))
short = lambda n: n.replace('void ', '').replace('(anonymous namespace)::', '')[:60]   # noqa: E731
for g, before, after in sorted(gaps, reverse=True)[:25]:
    print('%8.1f us   after %-60s before %s' % (g / 1e3, short(before), short(after)))
hist = [0, 0, 0, 0]
for g, _, _ in gaps:
    hist[0 if g < 2000 else 1 if g < 5000 else 2 if g < 20000 else 3] += g
print('idle by gap length: <2us %.2f ms, 2-5us %.2f ms, 5-20us %.2f ms, >20us %.2f ms' % tuple(h / 1e6 for h in hist))
