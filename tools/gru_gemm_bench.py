"""Short-K GEMM anatomy: M=7936, N=1536, K swept -> fixed cost per launch vs cost per K-tile; and the 256-row node form."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgg_amd import ops
from tools.gemm_bench import timeit
dev = 'cuda:0'
for M in (7936, 256):
    for K in (64, 512, 4096):
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(1536, K, device=dev) / 22).bfloat16()
        b = torch.randn(1536, device=dev)
        out = torch.empty(M, 1536, device=dev, dtype=torch.float32)
        ms = timeit(lambda: ops.gemm(A, W, b, out=out, out_dtype=torch.float32), reps=50)
        print('M=%5d K=%5d  %7.1f us  %6.1f TFLOP/s' % (M, K, ms * 1e3, 2.0 * M * 1536 * K / ms / 1e9))
