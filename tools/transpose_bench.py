"""Times sgg_transpose on the two fc6 weight-gradient operands.  usage: python tools/transpose_bench.py"""
import sys
import torch
sys.path.insert(0, '.')
from sgg_amd import ops

def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n

E = 7936
ef = torch.randn(E, 25088, device='cuda').bfloat16()
rect = torch.randn(E, 512, device='cuda')
d6 = torch.randn(E, 4096, device='cuda').bfloat16()
for name, fn, nbytes in (('ef+add', lambda: ops.transpose(ef, add=rect, group=49), 2 * ef.numel() * 2),
                         ('ef', lambda: ops.transpose(ef), 2 * ef.numel() * 2),
                         ('d6+colsum', lambda: ops.transpose(d6, want_colsum=True), 2 * d6.numel() * 2),
                         ('d6', lambda: ops.transpose(d6), 2 * d6.numel() * 2)):
    ms = timeit(fn)
    print('%-10s %.3f ms  %.2f TB/s' % (name, ms, nbytes / ms / 1e9))
