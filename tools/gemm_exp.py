"""Per-tile overhead of the 256x256 ping-pong GEMM: time against K at a fixed tile count (the intercept is prologue + epilogue + workgroup
turnover), and against the number of rounds at a tiny K.  `python tools/gemm_exp.py`"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgg_amd import ops
from tools.gemm_bench import timeit
dev = 'cuda:0'
dt = torch.float16


def run(M, N, K, out_dt=torch.bfloat16):
    A = torch.randn(M, K, device=dev).to(dt)
    W = torch.randn(N, K, device=dev).to(dt)
    out = torch.empty(M, N, device=dev, dtype=out_dt)
    ms = timeit(lambda: ops.gemm(A, W, out=out), reps=20)
    tiles = (M // 256) * (N // 256)
    print('M=%5d N=%5d K=%5d out=%-8s tiles %4d (%.2f rounds): %.3f ms %5.0f TF, %.1f us per round' % (
        M, N, K, str(out_dt)[6:], tiles, tiles / 256, ms, 2.0 * M * N * K / ms / 1e9, ms * 1e3 / max(1, -(-tiles // 256))), flush=True)


for K in (128, 256, 512, 1024, 1984, 3968):
    run(4096, 24576, K)
for N in (4096, 8192, 16384):
    run(4096, N, 128)
run(4096, 24576, 128, torch.float32)
run(4096, 24576, 128, torch.float16)


def run_gadd(M, N, K):
    from sgg_amd import _lib
    A = torch.randn(M, K, device=dev).to(dt)
    W = torch.randn(N, K, device=dev).to(dt)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    r = torch.randn(M, N // 49 + 1, device=dev)
    f = lambda: _lib.call('sgg_gemm_groupadd', ops._p(A), A.stride(0), ops._p(W), W.stride(0), ops._p(r), r.stride(0), 49, 0, ops._p(out), out.stride(0),
                          M, N, K, ops.dt(A), ops.dt(out), ops._stream())
    ms = timeit(f, reps=20)
    ref = (A[:64].float() @ W[:490].float().t()) + r[:64, :10].repeat_interleave(49, 1)
    err = float((out[:64, :490].float() - ref).abs().max() / ref.abs().max())
    print('group-add epilogue M=%d N=%d K=%d: %.3f ms %5.0f TF, rel err %.1e' % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9, err), flush=True)


run_gadd(4096, 24576, 3968)
run(4096, 24576, 3968)
