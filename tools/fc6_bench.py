"""fc6 forward / weight-gradient GEMMs on relu-sparse data (what the model feeds them)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgg_amd import ops
from tools.gemm_bench import timeit
dev = 'cuda:0'
g = torch.Generator().manual_seed(0)
A = torch.randn(7936, 25088, generator=g).relu().to(dev).bfloat16()
W = (torch.randn(4096, 25088, generator=g) / 160).to(dev).bfloat16()
out = torch.empty(7936, 4096, device=dev, dtype=torch.bfloat16)
ms = timeit(lambda: ops.gemm(A, W, out=out), reps=10)
print('fc6 fwd  %.3f ms  %.0f TF' % (ms, 2.0 * 7936 * 4096 * 25088 / ms / 1e9))
At = (torch.randn(4096, 7936, generator=g) / 50).to(dev).bfloat16()
dW = torch.empty(4096, 25088, device=dev, dtype=torch.float32)
Bt = A.t().contiguous()
ms = timeit(lambda: ops.gemm(At, Bt, out_dtype=torch.float32, out=dW), reps=10)
print('fc6 dW   %.3f ms  %.0f TF' % (ms, 2.0 * 7936 * 4096 * 25088 / ms / 1e9))
