"""How busy is the GPU inside one train step?  From a rocprofv3 --kernel-trace database (rocpd schema): the kernels of the last full steps,
their union coverage of the wall time (any stream), the time during which exactly one / two or more kernels run, and the largest gaps.
    rocprofv3 --kernel-trace -d out -o t -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32
    python tools/timeline.py out/.../t_results.db [first step]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
rows = db.execute('select %s, start, end%s from kernels order by start' % (name_col, (', ' + qcol) if qcol else '')).fetchall()
# step boundaries: the first VGG kernel of every step (the fused first block)
marks = [r[1] for r in rows if 'conv3x3_spatial_kernel' in r[0] and 'true' in r[0].split('conv3x3_spatial_kernel')[1][:40]]
if len(marks) < 4:
    marks = [r[1] for r in rows if 'image_prep_batch' in r[0]]
print('steps seen:', len(marks))
k0 = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) - 4     # first of the three whole steps looked at (default: the last three)
lo, hi = marks[k0], marks[k0 + 3]
sel = [r for r in rows if lo <= r[1] < hi]
wall = (hi - lo) / 1e3
events = []
for r in sel:
    events.append((r[1], 1))
    events.append((min(r[2], hi), -1))
events.sort()
busy1 = busy2 = 0.0
depth, last = 0, lo
gaps = []
for t, d in events:
    if depth == 0 and t - last > 0:
        gaps.append((t - last, last))
    if depth == 1:
        busy1 += t - last
    elif depth >= 2:
        busy2 += t - last
    depth += d
    last = t
print('3 steps: wall %.1f us per step; one kernel running %.1f us, two or more %.1f us, idle %.1f us per step' % (
    wall / 3, busy1 / 3e3, busy2 / 3e3, (wall * 1e3 - busy1 - busy2) / 3e3))
gaps.sort(reverse=True)
ends = sorted(sel, key=lambda r: r[2])
for g_, at in gaps[:9]:
    before = [r for r in ends if r[2] <= at + 1][-1:]
    after = [r for r in sel if r[1] >= at + g_ - 1][:1]
    print('  gap %.1f us after %s -> before %s' % (g_ / 1e3, before[0][0][:70] if before else '?', after[0][0][:70] if after else '?'))
print('idle gaps per step: %d; the ten largest (us): %s' % (len(gaps) / 3, ', '.join('%.1f' % (g / 1e3) for g, _ in gaps[:10])))
if qcol:
    per = {}
    for r in sel:
        per.setdefault(r[3], 0.0)
        per[r[3]] += (min(r[2], hi) - r[1]) / 3e3
    print('kernel time per queue and step (us):', {k: round(v, 1) for k, v in sorted(per.items(), key=lambda kv: -kv[1])})
    # what each queue spends its time on (us per step, launches per step)
    import re
    for qid in sorted(per, key=lambda k: -per[k]):
        agg = {}
        for r in sel:
            if r[3] != qid:
                continue
            nm = re.sub(r'\(anonymous namespace\)::|void |_ZN12_GLOBAL__N_1\d+', '', r[0])[:60]
            a = agg.setdefault(nm, [0.0, 0])
            a[0] += (min(r[2], hi) - r[1]) / 3e3
            a[1] += 1
        print('queue %s:' % qid)
        for nm, (us, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:16]:
            print('   %8.1f us  %5.1f x  %s' % (us, n / 3.0, nm))
import os
if os.environ.get('DUMP'):            # one step, every launch: start offset, duration, queue, name  (DUMP=<file>)
    lo1, hi1 = marks[k0], marks[k0 + 1]
    with open(os.environ['DUMP'], 'w') as f:
        for r in rows:
            if lo1 <= r[1] < hi1:
                f.write('%9.1f %8.1f  q%s  %s\n' % ((r[1] - lo1) / 1e3, (r[2] - r[1]) / 1e3, r[3] if qcol else '-', r[0][:110]))
