#!/bin/bash
# PMC passes over the sliced IMP step kernels (run on the GPU box from the repo root): tools/exp/pmc_imp.sh OUTDIR B
out=${1:-gpurun_out/pmc_imp}; B=${2:-128}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for form in d 0; do
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES -f csv -d $R/$out -o p1_$form -- python3 $R/tools/exp/pmc_imp.py $form $B > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -f csv -d $R/$out -o p2_$form -- python3 $R/tools/exp/pmc_imp.py $form $B > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_SENDMSG -f csv -d $R/$out -o p3_$form -- python3 $R/tools/exp/pmc_imp.py $form $B > /dev/null 2>&1
done
cd $R
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$out/**/*counter_collection.csv', recursive=True)):
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    for r in rows:
        k = r['Kernel_Name']
        if 'imp_' in k and ('dma' in k or 'sliced' in k or 'stream' in k):
            agg[(k.split('<')[0][-24:], r['Counter_Name'])].append(float(r['Counter_Value']))
    print(f.split('/')[-1])
    for (k, c), v in sorted(agg.items()):
        print('   %-26s %-24s %14.0f  (n=%d)' % (k, c, sum(v) / len(v), len(v)))
PY
