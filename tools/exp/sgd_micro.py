"""SGD update kernel alone on fc6-sized tensors (102.8 M parameters): us and GB/s per workgroup count (20 bytes per parameter:
bf16 gradient in; f32 parameter and momentum in and out; bf16 shadow out)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sgg_amd import _lib, ops  # noqa: E402

dev = 'cuda:0'
n = 25088 * 4096
p = torch.randn(n, device=dev)
buf = torch.zeros(n, device=dev)
g = torch.randn(n, device=dev).bfloat16()
sh = torch.empty(n, device=dev, dtype=torch.bfloat16)
norm = torch.ones(1, device=dev)
arr = lambda xs, dt: np.ascontiguousarray(np.array(xs, dtype=dt))  # noqa: E731
pp, gp, bp, sp = (arr([t.data_ptr()], np.uint64) for t in (p, g, buf, sh))
nn, lr = arr([n], np.int64), arr([1e-4], np.float32)
for blocks in [int(b) for b in sys.argv[1:]] or [256, 512, 1024, 2048, 4096, 8192, 25088]:
    def run():
        _lib.call('sgg_sgd_multi', pp.ctypes.data, gp.ctypes.data, bp.ctypes.data, sp.ctypes.data, nn.ctypes.data, lr.ctypes.data, 1,
                  1e-4, 0.9, 0, norm.data_ptr(), 5.0, 1.0, ops.dt(g), blocks, ops._stream())
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print('blocks %6d: %7.1f us  %5.0f GB/s' % (blocks, ms * 1e3, 20.0 * n / ms / 1e6), flush=True)
