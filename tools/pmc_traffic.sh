#!/bin/bash
# HBM-side traffic of the roofline kernels (fc6 GEMMs, IMP step) from two rocprofv3 PMC passes, one counter each, and the kernel-trace
# summaries of the train / inference bench.  Run on the GPU box from the repo root:   bash tools/pmc_traffic.sh gpurun_out/pmc_r04 r04
out=${1:-gpurun_out/pmc_r04}
tag=${2:-r04}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/$out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -f csv -d $R/$out -o $c -- python3 $R/tools/pmc_kernels.py > $R/$out/$c.log 2>&1
done
cd $R
python3 - <<PY
import collections, csv, glob, json
res = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    files = glob.glob('$out/**/%s_counter_collection.csv' % c, recursive=True)
    if not files:
        continue
    agg = collections.defaultdict(list)
    rows = [r for r in csv.DictReader(open(files[0])) if r['Counter_Name'] == c]
    rows.sort(key=lambda r: int(r.get('Dispatch_Id', 0) or 0))
    for r in rows:
        wgs = int(r['Grid_Size']) // max(int(r['Workgroup_Size']), 1)
        agg[(r['Kernel_Name'], wgs)].append(float(r['Counter_Value']))
    # the f16 fc6 launch and, since round 6, the x3 launch on pair operands share kernel and grid: tools/pmc_kernels.py runs three of the first,
    # then three of the second
    for (k, wgs) in list(agg):
        if 'mfma_pingpong_kernel' in k and wgs == 256 and len(agg[(k, wgs)]) == 6:
            v = agg[(k, wgs)]
            agg[(k, wgs)] = v[:3]
            agg[(k + ' x3', wgs)] = v[3:]
    for (k, wgs), v in agg.items():
        key = None
        if 'mfma_pingpong_kernel' in k and wgs == 256:
            key = 'fc6_edge_gemm_x3' if k.endswith(' x3') else 'fc6_edge_gemm'
        elif 'mfma_pingpong_kernel' in k and wgs == 512:
            key = 'box_fc6_gemm'
        elif 'mfma_pingpong_kernel' in k and wgs >= 1500:
            key = 'fc6_dW_gemm'
        elif 'imp_ctx_sliced_kernel' in k:
            key = 'imp_ctx_B8' if wgs <= 256 else 'imp_ctx_B128'
        elif 'gru_gate_proj_kernel' in k:
            key = 'gate_proj_B8' if wgs <= 4096 else 'gate_proj_B128'
        if key:
            res.setdefault(key, {})[c + '_KiB_avg'] = sum(v) / len(v)
            res[key]['launches'] = len(v)
for key, d in res.items():
    # MI355X_MICROARCH.md (HBM / rocprofv3 section): counters in KiB; FETCH_SIZE reports half of the bytes of wide coalesced reads on gfx950
    d['traffic_bytes'] = int(1024 * (2 * d.get('FETCH_SIZE_KiB_avg', 0) + d.get('WRITE_SIZE_KiB_avg', 0)))
res['_about'] = ('HBM-side traffic per launch of the roofline kernels from rocprofv3 PMC passes on MI355X ($tag, f16): '
                 'rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -f csv -- python3 tools/pmc_kernels.py, one counter per pass '
                 '(tools/pmc_traffic.sh). Units KiB; FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md; the counters sit '
                 'on the fabric side of L2, Infinity-Cache hits included.')
json.dump(res, open('$out/pmc_$tag.json', 'w'), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
