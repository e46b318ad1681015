#!/usr/bin/env python3
"""Runs the roofline kernels in isolation (for rocprofv3 --pmc passes): the fc6 GEMM on the unordered box pairs (forward and weight
gradient, B=8: 3968 pairs), the IMP gather / gate / scatter launch and the projected edge-GRU gate kernel at B=8 and B=128 (f16).  Usage under the profiler:
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d out -o fetch -- python3 tools/pmc_kernels.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d out -o write -- python3 tools/pmc_kernels.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgg_amd import ops  # noqa: E402

dev = 'cuda:0'
g = torch.Generator().manual_seed(0)
dt = torch.float16
# fc6 on the unordered box pairs (sgg_amd/pairing.py): [U, 25088] . [4096, 25088]^T, f32 out
M, N, K1 = 3968, 4096, 25088
A = torch.randn(M, K1, generator=g).to(dev).to(dt).relu()
W = (torch.randn(N, K1, generator=g) / 160).to(dev).to(dt)
out = torch.empty(M, N, device=dev, dtype=torch.float32)
for _ in range(3):
    ops.gemm(A, W, out=out, out_dtype=torch.float32)
torch.cuda.synchronize()
# the SGDet box head's fc6 at BASELINE configs[2]'s size: [8000 proposals, 25088] . [4096, 25088]^T, 16-bit out (512 workgroups = two rounds)
Ab = torch.randn(8000, K1, generator=g).to(dev).to(dt).relu()
bb = torch.randn(N, generator=g).to(dev)
for _ in range(3):
    ops.gemm(Ab, W, bb, ops.ACT_RELU)
torch.cuda.synchronize()
del Ab
# the x3 mode's form of the first contraction: the same fc6 on PAIR operands (SGG_PAIR16: the MFMA loop walks hi.hi + hi.lo + lo.hi)
A32, W32 = A.float(), W.float()
ops.set_split3(True)
for _ in range(3):
    ops.gemm(A32, W32, out=out, out_dtype=torch.float32)
ops.set_split3(False)
torch.cuda.synchronize()
del A32, W32
# its weight gradient (the train step's largest contraction): dW[4096, 25088] = (pair-summed d_pre6 [U, 4096])^T . pooled [U, 25088], the
# TN form of the ping-pong kernel: both operands as they lie
Mg, Ng, Kg = 4096, 25088, 3968
At = (torch.randn(Kg, Mg, generator=g) / 50).to(dev).to(dt)
Bt = torch.randn(Kg, Ng, generator=g).to(dev).to(dt).relu()
for _ in range(3):
    dW = ops.gemm_tn_full_waves(At, Bt, out_dtype=dt)
torch.cuda.synchronize()
del At, Bt, dW
# the IMP step's launch (sgg_imp_ctx_fwd: every edge row read once, two gated sums per node) and the gate kernel that takes the place
# of the edge inputs (sgg_gru_gate_proj_fwd), at B = 8 and B = 128
H, n = 512, 32
gw = (torch.randn(4, 2 * H, generator=g) / 30).to(dev)
gb = torch.randn(4, generator=g).to(dev)
b_ih = torch.randn(3 * H, generator=g).to(dev)
for B in (8, 128):
    Nn, E = n * B, n * (n - 1) * B
    im = torch.arange(B, device=dev).repeat_interleave(n)
    rel, _ = ops.pair_index_eval(im)
    rel = rel[:E]
    csr = ops.edge_csr(rel, Nn, im, graphs=(B, n, n * (n - 1)))
    v = torch.randn(Nn, H, generator=g).to(dev).to(dt)
    e = torch.randn(E, H, generator=g).to(dev).to(dt)
    nd, ed = (v.float() @ gw[:, :H].t()).contiguous(), (e.float() @ gw[:, H:].t()).contiguous()
    gh, P = torch.randn(E, 3 * H, generator=g).to(dev).to(ops.gh_dtype(dt)), torch.randn(Nn, 3 * H, generator=g).to(dev)
    for _ in range(3):
        ops.imp_ctx(e, csr, Nn, nd, ed, gb)
    for _ in range(3):
        ops.gru_gate_proj(gh, P, b_ih, csr, nd, ed, gb, e, dot_w=gw[:, H:])
    torch.cuda.synchronize()
print('done')
