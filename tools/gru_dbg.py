import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgg_amd import ops
from tools.gemm_bench import timeit
dev = 'cuda:0'
M, K = 7936, 64
A = torch.randn(M, K, device=dev).bfloat16(); W = (torch.randn(1536, K, device=dev) / 22).bfloat16(); b = torch.randn(1536, device=dev)
out = torch.empty(M, 1536, device=dev, dtype=torch.float32)
print(os.environ.get('SGG_GEMM_DBG'), os.environ.get('SGG_GEMM_FORCE'), '%.1f us' % (timeit(lambda: ops.gemm(A, W, b, out=out, out_dtype=torch.float32), reps=50) * 1e3))
