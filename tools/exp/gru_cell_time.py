"""What one edge-GRU call costs today (two GEMMs with f32 outputs + the gate kernel) against the GEMM part of a fused cell
(one launch, K = [x | h], N = 4 blocks of H columns: r, z, n_x, n_h)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tools'))
from gemm_bench import timeit  # noqa: E402
from sgg_amd import ops  # noqa: E402

dev, dt, H = 'cuda:0', torch.bfloat16, 512
for M in (7936, 256):
    x = torch.randn(M, H, device=dev).to(dt)
    h = torch.randn(M, H, device=dev).to(dt)
    wih = (torch.randn(3 * H, H, device=dev) / 22).to(dt)
    whh = (torch.randn(3 * H, H, device=dev) / 22).to(dt)
    bih, bhh = torch.randn(3 * H, device=dev), torch.randn(3 * H, device=dev)
    gi = torch.empty(M, 3 * H, device=dev, dtype=torch.float32)
    gh = torch.empty(M, 3 * H, device=dev, dtype=torch.float32)
    out = torch.empty(M, H, device=dev, dtype=dt)
    dot_w = torch.randn(4, H, device=dev)
    t_gi = timeit(lambda: ops.gemm(x, wih, bih, out=gi, out_dtype=torch.float32), reps=50)
    t_gate = timeit(lambda: ops.gru_gate(gi, gh, None, h, dt, out=out, dot_w=dot_w), reps=50)

    def cell():
        ops.gemm(x, wih, bih, out=gi, out_dtype=torch.float32)
        ops.gemm(h, whh, bhh, out=gh, out_dtype=torch.float32)
        ops.gru_gate(gi, gh, None, h, dt, out=out, dot_w=dot_w)
    t_cell = timeit(cell, reps=50)
    w1 = (torch.randn(4 * H, H, device=dev) / 22).to(dt)
    w2 = (torch.randn(4 * H, H, device=dev) / 22).to(dt)
    b4 = torch.randn(4 * H, device=dev)
    o4 = torch.empty(M, 4 * H, device=dev, dtype=dt)
    t_f = timeit(lambda: ops.gemm(x, w1, b4, A2=h, W2=w2, out=o4), reps=50)
    o3 = torch.empty(M, 3 * H, device=dev, dtype=dt)
    t_b = timeit(lambda: ops.gemm(x, wih, bih, out=o3), reps=50)
    print('M=%5d: one GEMM (f32 out) %.1f us, gate kernel %.1f us, whole cell %.1f us | fused-shape GEMM [M,1024]x[2048,1024] bf16 out %.1f us | '
          'one GEMM with bf16 out %.1f us' % (M, t_gi * 1e3, t_gate * 1e3, t_cell * 1e3, t_f * 1e3, t_b * 1e3))
