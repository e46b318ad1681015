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
python3 bench.py --mode gqa_gan --steps 10 --warmup 3 > $out/${tag}_bench_gqa_gan.json 2> $out/bench_gqa_gan.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_train_driver_invocation.json 2> $out/bench_train_driver.err
cd /tmp && export TMPDIR=/tmp
for mode in train infer sgdet gqa_gan; do
  extra="--no-side-modes"; steps=10
  [ $mode = gqa_gan ] && steps=3
  timeout 900 rocprofv3 --kernel-trace -d $out -o $mode -- python3 $R/bench.py --mode $mode --steps $steps --warmup 3 --no-cpu-baseline --no-f32 $extra > $out/prof_$mode.log 2>&1
  db=$(find $out -name "${mode}_results.db" | head -1)
  name=$mode; [ $mode = gqa_gan ] && name=gan
  [ -n "$db" ] && python3 $R/tools/kernel_stats.py $db > $out/${tag}_${name}_kernel_stats.txt
done
# the x3 mode (the parity-qualified throughput): the same workload through tools/mode_steps.py
for what in infer train; do
  timeout 600 rocprofv3 --kernel-trace -d $out -o x3_$what -- python3 $R/tools/mode_steps.py x3 $what 10 > $out/prof_x3_$what.log 2>&1
  db=$(find $out -name "x3_${what}_results.db" | head -1)
  [ -n "$db" ] && python3 $R/tools/kernel_stats.py $db > $out/${tag}_x3_${what}_kernel_stats.txt
done
cd $R
bash tools/pmc_traffic.sh gpurun_out/prof_$tag/pmc $tag > $out/pmc.log 2>&1
cp gpurun_out/prof_$tag/pmc/pmc_$tag.json $out/pmc_${tag}.json 2>/dev/null
find $out -name "*.db" -delete; find $out -name "*.csv" -size +2M -delete
ls -la $out
