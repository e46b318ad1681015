#!/usr/bin/env python3
"""Micro-benchmark of sgg_gemm / sgg_conv3x3_relu at the BASELINE shapes (random data, HIP-event timed)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgg_amd import ops  # noqa: E402


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = 'cuda:0'
    dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
    shapes = [('fc6_edge', 7936, 4096, 25600), ('fc7_edge', 7936, 4096, 4096), ('unary', 7936, 512, 4096),
              ('gru', 7936, 1536, 512), ('fc6_obj', 256, 4096, 25088), ('rel_fc', 7936, 51, 512),
              ('sq4096', 4096, 4096, 4096), ('sq8192', 8192, 8192, 8192), ('fc6_8192', 8192, 4096, 25600)]
    for name, M, N, K in ([] if os.environ.get('CONV_ONLY') else shapes):
        A = torch.randn(M, K, device=dev).to(dt)
        W = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
        b = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=dt)
        ms = timeit(lambda: ops.gemm(A, W, b, ops.ACT_RELU, out=out))
        print('%-10s M=%6d N=%5d K=%6d  %8.3f ms  %7.1f TFLOP/s' % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
    B = 8
    for name, H, Cin, Cout in [('conv1_2', 592, 64, 64), ('conv2_1', 296, 64, 128), ('conv2_2', 296, 128, 128),
                               ('conv3_1', 148, 128, 256), ('conv3_2', 148, 256, 256), ('conv4_1', 74, 256, 512),
                               ('conv4_2', 74, 512, 512), ('conv5_1', 37, 512, 512)]:
        x = torch.randn(B, H + 2, H + 2, Cin, device=dev).to(dt)
        w = (torch.randn(Cout, 3, 3, Cin, device=dev) / (3 * Cin ** 0.5)).to(dt)
        b = torch.randn(Cout, device=dev)
        y = torch.empty(B, H + 2, H + 2, Cout, device=dev, dtype=dt)
        ms = timeit(lambda: ops.conv3x3_relu(x, w, b, y, 1))
        print('%-10s H=%4d Cin=%4d Cout=%4d  %8.3f ms  %7.1f TFLOP/s' % (name, H, Cin, Cout, ms, 2.0 * B * H * H * Cout * 9 * Cin / ms / 1e9))


if __name__ == '__main__':
    main()
