#!/usr/bin/env python3
"""The 12 MFMA convolutions of VGG-16 at the detector's shapes (8 x 608 x 608 after padding), as the detector launches them (pool fused where
it fuses it), each under the routing the library picks and under both forced kernels (SGG_CONV_FORCE=s|g): us and TFLOP/s per layer."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgg_amd import ops  # noqa: E402
from tools.gemm_bench import timeit  # noqa: E402

dev = 'cuda:0'
dt = {'f16': torch.float16, 'bf16': torch.bfloat16, 'f32': torch.float32}[sys.argv[1] if len(sys.argv) > 1 else 'f16']
B = int(os.environ.get('B', '8'))
layers = [('conv1_2', 608, 64, 64, True), ('conv2_1', 304, 64, 128, False), ('conv2_2', 304, 128, 128, True), ('conv3_1', 152, 128, 256, False),
          ('conv3_2', 152, 256, 256, False), ('conv3_3', 152, 256, 256, True), ('conv4_1', 76, 256, 512, False), ('conv4_2', 76, 512, 512, False),
          ('conv4_3', 76, 512, 512, True), ('conv5_1', 38, 512, 512, False), ('conv5_2', 38, 512, 512, False), ('conv5_3', 38, 512, 512, False)]
tot = {}
for name, H, Cin, Cout, pool in layers:
    x = torch.randn(B, H + 2, H + 2, Cin, device=dev).to(dt)
    x[:, 0] = 0; x[:, -1] = 0; x[:, :, 0] = 0; x[:, :, -1] = 0
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) / (3 * Cin ** 0.5)).to(dt)
    b = torch.randn(Cout, device=dev)
    Ho = H // 2 if pool else H
    y = torch.zeros(B, Ho + 2, Ho + 2, Cout, device=dev, dtype=dt)
    flop = 2.0 * B * H * H * Cout * 9 * Cin
    row = []
    for force in (None, 's', 'g'):
        if force is None:
            os.environ.pop('SGG_CONV_FORCE', None)
        else:
            os.environ['SGG_CONV_FORCE'] = force
        if force == 'g' and pool:
            row.append('      -      ')
            continue
        try:
            ms = timeit(lambda: ops.conv3x3_relu(x, w, b, y, 1, pool=pool), reps=20)
            row.append('%7.1f us %5.0f' % (ms * 1e3, flop / ms / 1e9))
            tot[force] = tot.get(force, 0.0) + (ms if not (force == 'g' and pool) else 0)
        except Exception as e:
            row.append('   %s   ' % type(e).__name__)
    print('%-8s H=%3d %3d->%3d pool=%d | routed %s | spatial %s | gemm %s' % (name, H, Cin, Cout, pool, row[0], row[1], row[2]), flush=True)
os.environ.pop('SGG_CONV_FORCE', None)
print('routed total %.3f ms = %.0f TFLOP/s (12 layers, %.1f GFLOP)' % (tot[None], (226.13e9 * B - 2.0 * B * 608 * 608 * 64 * 27) / tot[None] / 1e9, 226.13 * B))
