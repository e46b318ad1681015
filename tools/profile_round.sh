#!/bin/bash
# Everything kept under profiles/ for one round, made on the GPU box from the repo root:   bash tools/profile_round.sh r04
#   <tag>_bench_train.json / _bench_infer.json        the bench lines (bench.py, default arguments / --mode infer)
#   <tag>_train_kernel_stats.txt / _infer_...          rocprofv3 --kernel-trace of the same bench commands (short runs), per-kernel summary
#   pmc_<tag>.json                                     FETCH_SIZE / WRITE_SIZE passes of the roofline kernels (tools/pmc_traffic.sh)
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/prof_$tag
mkdir -p $out
cd $R
python3 bench.py > $out/${tag}_bench_train.json 2> $out/bench_train.err
python3 bench.py --mode infer --no-cpu-baseline --no-f32 > $out/${tag}_bench_infer.json 2> $out/bench_infer.err
python3 bench.py --mode sgdet --steps 50 > $out/${tag}_bench_sgdet.json 2> $out/bench_sgdet.err
cd /tmp && export TMPDIR=/tmp
for mode in train infer sgdet; do
  timeout 900 rocprofv3 --kernel-trace -d $out -o $mode -- python3 $R/bench.py --mode $mode --steps 10 --warmup 3 --no-cpu-baseline --no-f32 > $out/prof_$mode.log 2>&1
  db=$(find $out -name "${mode}_results.db" | head -1)
  [ -n "$db" ] && python3 $R/tools/kernel_stats.py $db > $out/${tag}_${mode}_kernel_stats.txt
done
cd $R
bash tools/pmc_traffic.sh gpurun_out/prof_$tag/pmc $tag > $out/pmc.log 2>&1
cp gpurun_out/prof_$tag/pmc/pmc_$tag.json $out/pmc_${tag}.json 2>/dev/null
find $out -name "*.db" -delete; find $out -name "*.csv" -size +2M -delete
ls -la $out
