"""dW = dY^T X: TN kernel (no transposed copies) vs transposes + NT GEMM, at the train step's shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgg_amd import ops
from tools.gemm_bench import timeit
dev = 'cuda:0'
for name, Mred, N, K in (('gru_w_ih', 31744, 1536, 512), ('fc7', 7936, 4096, 4096), ('unary', 7936, 512, 4096),
                         ('fc6', 7936, 4096, 25088), ('fc6_pairs', 3968, 4096, 25088), ('fc6_obj', 256, 4096, 25088), ('rect1', 31744, 256, 128)):
    A = (torch.randn(Mred, N, device=dev) / 8).bfloat16()
    B = torch.randn(Mred, K, device=dev).relu().bfloat16()
    out = torch.empty(N, K, device=dev, dtype=torch.float32)
    t_tn = timeit(lambda: ops.gemm_tn(A, B, out=out), reps=20)
    extra = ''
    if N % 256 == 0 and K % 256 == 0 and (N // 256) * (K // 256) < 128:
        t256 = (N // 256) * (K // 256)
        for sp in (max(2, 128 // t256), max(2, 256 // t256), max(2, 512 // t256)):
            if Mred // 32 // sp >= 8 and sp <= Mred // 64:
                t_s = timeit(lambda: ops.gemm_tn(A, B, out=out, splits=sp), reps=20)
                extra += '  splits=%d %.1f us' % (sp, t_s * 1e3)
    t_nt = timeit(lambda: ops.gemm(ops.transpose(A), ops.transpose(B), out_dtype=torch.float32, out=out), reps=20)
    fl = 2.0 * Mred * N * K
    print('%-8s Mred=%5d N=%4d K=%5d   TN %8.1f us (%6.0f TF)   transposes+NT %8.1f us (%6.0f TF)' % (name, Mred, N, K, t_tn * 1e3, fl / t_tn / 1e9, t_nt * 1e3, fl / t_nt / 1e9) + extra)
