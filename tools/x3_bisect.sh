#!/bin/bash
# which test file, run before the x3 bench-size tests in the same process, makes them fail?  (one pytest process per candidate)
mkdir -p gpurun_out/x3_bisect
for f in blob dense determinism dist evaluate f16 freq_bias fuzz gan_dist gan_model gan_ops kernels model pairing; do
  timeout 900 python -m pytest tests/test_${f}_gpu.py tests/test_zz_x3_bench_gpu.py -q -m gpu > gpurun_out/x3_bisect/$f.log 2>&1
  echo "$f rc=$? $(grep -E 'passed|failed' gpurun_out/x3_bisect/$f.log | tail -1)"
  cp gpurun_out/x3_stage_diag.json gpurun_out/x3_bisect/$f.diag.json 2>/dev/null && rm -f gpurun_out/x3_stage_diag.json
  cp gpurun_out/r05_parity_bench_config.json gpurun_out/x3_bisect/$f.parity.json 2>/dev/null
done
