"""the fc6 weight-gradient operand transposes of a B=8 train step: X^T with the rect term added (7936 x 25088, group 49) and dY^T."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sgg_amd import ops
dev = 'cuda:0'
def timeit(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for R, C, grp in ((7936, 25088, 49), (7936, 25088, 0), (7936, 4096, 0)):
    x = torch.randn(R, C, device=dev).to(torch.bfloat16)
    add = torch.randn(R, C // grp, device=dev) if grp else None
    t = timeit(lambda: ops.transpose(x, add=add, group=grp or 1))
    print('%d x %d  add=%s  %.1f us  %.2f TB/s' % (R, C, bool(grp), t * 1e3, 2 * R * C * 2 / t / 1e9))
