"""fc6 on the 256 object rows: [256 x 25088] . [4096 x 25088]^T -- split count of the split-K launch"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tools'))
from gemm_bench import timeit  # noqa: E402
from sgg_amd import ops  # noqa: E402

dev, dt = 'cuda:0', torch.bfloat16
A = torch.randn(256, 25088, device=dev).to(dt)
W = (torch.randn(4096, 25088, device=dev) / 160).to(dt)
b = torch.randn(4096, device=dev)
for sp in (None, 4, 6, 8, 12, 16, 24, 32):
    t = timeit(lambda: ops.gemm(A, W, b, ops.ACT_RELU, splits=sp), reps=30)
    print('splits %s: %.1f us  (%.2f TB/s of weights)' % (sp, t * 1e3, W.numel() * 2 / t / 1e9))
