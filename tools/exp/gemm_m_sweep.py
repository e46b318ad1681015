import sys, torch
sys.path.insert(0, '.')
from sgg_amd import ops
from tools.gemm_bench import timeit
dev, dt = 'cuda:0', torch.bfloat16
for K in (25600, 4096):
    for M in (7424, 7680, 7936, 8192):
        N = 4096
        A = torch.randn(M, K, device=dev).to(dt); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
        b = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev, dtype=dt)
        ms = min(timeit(lambda: ops.gemm(A, W, b, ops.ACT_RELU, out=out), reps=20) for _ in range(3))
        tiles = (M // 256) * (N // 256)
        print('M=%5d K=%5d tiles=%4d  %7.3f ms %7.1f TF  per-tile-round %.3f ms' % (M, K, tiles, ms, 2.0 * M * N * K / ms / 1e9, ms / -(-tiles // 256)))
