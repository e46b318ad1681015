"""Short-K GEMM anatomy (run under rocprofv3 --kernel-trace): fixed cost of the tile kernels vs tile count."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgg_amd import ops
dev = 'cuda:0'
for (M, N, K) in ((256, 256, 64), (256, 256, 512), (2048, 1536, 64), (7936, 1536, 64), (7936, 1536, 512), (7936, 4096, 64)):
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) / 22).bfloat16()
    b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.float32)
    for _ in range(20):
        ops.gemm(A, W, b, out=out, out_dtype=torch.float32)
    torch.cuda.synchronize()
    print(M, N, K)
