#!/bin/bash
# Counter evidence for the RoIAlign forward: L2 hit rate and where the waves wait, from separate rocprofv3 PMC passes of
# tools/roi_bench.py (the union-box launch: 3968 workgroups).  bash tools/roi_pmc.sh > gpurun_out/r06/roi_pmc.txt
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/roi_pmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for set in "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $set -f csv -d $out -o $name -- python3 $R/tools/roi_bench.py > $out/$name.log 2>&1
done
cd $R
python3 - <<PY
import collections, csv, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$out/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'roi_align_kernel' not in r['Kernel_Name']:
            continue
        wgs = int(r['Grid_Size']) // max(int(r['Workgroup_Size']), 1)
        dtype = 'f32' if '<float>' in r['Kernel_Name'] else ('f16' if 'Float16' in r['Kernel_Name'] or 'DF16' in r['Kernel_Name'] or '_Float16' in r['Kernel_Name'] else 'bf16')
        agg[(dtype, wgs)][r['Counter_Name']].append(float(r['Counter_Value']))
for key in sorted(agg):
    print('roi_align_kernel %s, %d workgroups:' % key)
    c = {k: sum(v) / len(v) for k, v in agg[key].items()}
    for k in sorted(c):
        print('   %-34s %16.1f  (avg of %d launches)' % (k, c[k], len(agg[key][k])))
    if 'TCC_HIT_sum' in c and 'TCC_MISS_sum' in c:
        print('   L2 hit rate %.4f' % (c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum'])))
    if 'SQ_WAVE_CYCLES' in c:
        w = c['SQ_WAVE_CYCLES']
        print('   of the wave cycles: parked (s_waitcnt / barrier) %.3f, issue-stalled %.3f (of which LDS %.3f), issuing %.3f' % (
            c.get('SQ_WAIT_ANY', 0) / w, c.get('SQ_WAIT_INST_ANY', 0) / w, c.get('SQ_WAIT_INST_LDS', 0) / w, c.get('SQ_ACTIVE_INST_ANY', 0) / w))
PY
