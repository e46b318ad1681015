#!/usr/bin/env python3
"""One configuration of the VGG convolutions under the environment it is started with (SGG_CONV_FORCE / SGG_CONV_WIDE / SGG_HIP_LIB are read once
by the library): us and TFLOP/s per layer.  `python tools/conv_one.py [f16|bf16] [layer substring]`"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgg_amd import ops  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

dev = 'cuda:0'
dt = {'f16': torch.float16, 'bf16': torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else 'f16']
only = sys.argv[2] if len(sys.argv) > 2 else ''
B = int(os.environ.get('B', '8'))
layers = [('conv1_2', 608, 64, 64, True), ('conv2_1', 304, 64, 128, False), ('conv2_2', 304, 128, 128, True), ('conv3_1', 152, 128, 256, False),
          ('conv3_2', 152, 256, 256, False), ('conv3_3', 152, 256, 256, True), ('conv4_1', 76, 256, 512, False), ('conv4_2', 76, 512, 512, False),
          ('conv4_3', 76, 512, 512, True), ('conv5_1', 38, 512, 512, False)]
tot = 0.0
for name, H, Cin, Cout, pool in layers:
    if only not in name:
        continue
    x = torch.randn(B, H + 2, H + 2, Cin, device=dev).to(dt)
    x[:, 0] = 0; x[:, -1] = 0; x[:, :, 0] = 0; x[:, :, -1] = 0
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) / (3 * Cin ** 0.5)).to(dt)
    b = torch.randn(Cout, device=dev)
    Ho = H // 2 if pool else H
    y = torch.zeros(B, Ho + 2, Ho + 2, Cout, device=dev, dtype=dt)
    flop = 2.0 * B * H * H * Cout * 9 * Cin
    ms = timeit(lambda: ops.conv3x3_relu(x, w, b, y, 1, pool=pool), reps=20)
    tot += ms * (3 if name == 'conv5_1' else 1)
    # check against torch on a corner of the output (the routed kernel's numbers)
    # the whole LAST image against torch (edge tiles included), and that the output border stayed zero
    ref = torch.relu(torch.nn.functional.conv2d(x[-1:].permute(0, 3, 1, 2).float(), w.permute(0, 3, 1, 2).float(), b))
    if pool:
        ref = torch.nn.functional.max_pool2d(ref, 2)
    got = y[-1:, 1:-1, 1:-1].permute(0, 3, 1, 2).float()
    err = float((got - ref).abs().max() / ref.abs().max())
    if float(y[:, 0].abs().max()) or float(y[:, -1].abs().max()) or float(y[:, :, 0].abs().max()) or float(y[:, :, -1].abs().max()):
        err = float('nan')
    print('%-8s H=%3d %3d->%3d pool=%d | %7.1f us %5.0f TF | rel err %.1e' % (name, H, Cin, Cout, pool, ms * 1e3, flop / ms / 1e9, err), flush=True)
print('total (conv5 x3) %.3f ms' % tot)
