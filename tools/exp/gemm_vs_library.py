"""The two fc6 contractions of a B=8 train step: this repo's ping-pong MFMA kernel (through ops.gemm) next to torch.matmul (hipBLASLt /
rocBLAS as shipped with PyTorch-ROCm) on the same operands.  Context only: the product path never calls the library for these."""
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
for name, M, N, K in (('fc6 forward  [7936 x 25088] . [4096 x 25088]^T', 7936, 4096, 25088),
                      ('fc6 dW       [4096 x 7936] . [25088 x 7936]^T', 4096, 25088, 7936)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t_mine = timeit(lambda: ops.gemm(A, W, out=out))
    Wt = W.t()
    t_lib = timeit(lambda: torch.matmul(A, Wt, out=out))
    fl = 2.0 * M * N * K
    print('%s   this repo %.3f ms (%.0f TF)   torch.matmul %.3f ms (%.0f TF)' % (name, t_mine, fl / t_mine / 1e9, t_lib, fl / t_lib / 1e9))
