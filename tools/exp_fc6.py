import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgg_amd import _lib
if os.environ.get('SGG_LIB'):
    _lib.LIB_PATH = os.environ['SGG_LIB']
from sgg_amd import ops
from tools.gemm_bench import timeit
M, N, K = 7936, 4096, 25600
A = torch.randn(M, K, device='cuda').bfloat16(); W = (torch.randn(N, K, device='cuda') / 160).bfloat16()
out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
ms = timeit(lambda: ops.gemm(A, W, None, out=out))
print('V=%s  %.3f ms  %.1f TF' % (os.environ.get('SGG_PP_V', '0'), ms, 2.0 * M * N * K / ms / 1e9))
