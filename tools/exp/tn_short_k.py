"""fc6-on-nodes weight gradient: d_p6[256,4096]^T . nf[256,25088] -> [4096,25088]: the TN kernel against transposes + the NT kernels"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tools'))
from gemm_bench import timeit  # noqa: E402
from sgg_amd import ops  # noqa: E402

dev, dt = 'cuda:0', torch.bfloat16
for R in (256, 512):
    X = torch.randn(R, 4096, device=dev).to(dt)
    Y = torch.randn(R, 25088, device=dev).to(dt)
    for od in (torch.bfloat16, torch.float32):
        t_tn = timeit(lambda: ops.gemm_tn(X, Y, out_dtype=od), reps=30)
        t_nt = timeit(lambda: ops.gemm(ops.transpose(X), ops.transpose(Y), out_dtype=od), reps=30)
        xt, yt = ops.transpose(X), ops.transpose(Y)
        t_g = timeit(lambda: ops.gemm(xt, yt, out_dtype=od), reps=30)
        print('rows %d out %s: TN kernel %.1f us | transposes + NT %.1f us (NT alone %.1f)' % (R, str(od)[6:], t_tn * 1e3, t_nt * 1e3, t_g * 1e3))
