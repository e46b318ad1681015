"""conv + MaxPool2d(2): the spatial kernel with the pool in its epilogue against the implicit-GEMM kernel + the pool kernel (B=8, bf16)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tools'))
from gemm_bench import timeit  # noqa: E402
from sgg_amd import ops  # noqa: E402

dev, dt, B = 'cuda:0', torch.bfloat16, 8
for name, H, C in (('conv3_3', 148, 256), ('conv4_3', 74, 512), ('conv2_2', 296, 128)):
    x = torch.randn(B, H + 2, H + 2, C, device=dev).to(dt)
    w = (torch.randn(C, 3, 3, C, device=dev) / (3 * C ** 0.5)).to(dt)
    b = torch.randn(C, device=dev)
    yp = torch.zeros(B, H // 2 + 2, H // 2 + 2, C, device=dev, dtype=dt)
    yf = torch.zeros(B, H + 2, H + 2, C, device=dev, dtype=dt)
    t_fused = timeit(lambda: ops.conv3x3_relu(x, w, b, yp, 1, pool=True))

    def two():
        ops.conv3x3_relu(x, w, b, yf, 1)
        ops.maxpool2x2(yf, yp, 1)
    t_two = timeit(two)
    print('%s: pool fused in the patch kernel %.3f ms | conv (default routing) + pool kernel %.3f ms' % (name, t_fused, t_two), flush=True)
