"""Randomised parity sweep (GPU): many random shapes / graphs for the kernels with the most intricate indexing -- the sliced and the
node-centric IMP step, the GEMM family (tile, ping-pong, split-K, TN; K-split operands, activations) -- against dense torch fp32.
    python tools/fuzz_kernels.py [seconds]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sgg_oracle as O
from sgg_amd import ops
dev = 'cuda:0'
cu = lambda t: (torch.from_numpy(np.ascontiguousarray(t)) if isinstance(t, np.ndarray) else t.contiguous()).to(dev)
rng = np.random.RandomState(0)
stats = {'imp': 0, 'gemm': 0, 'tn': 0}


def fuzz_imp():
    H = int(rng.choice([32, 64, 128, 256, 512]))
    dtype = torch.bfloat16 if rng.rand() < 0.5 else torch.float32
    nb = int(rng.randint(1, 7))
    sizes = [int(rng.randint(1, 50 if H >= 128 else 20)) for _ in range(nb)]
    im = np.concatenate([np.full(n, b) for b, n in enumerate(sizes)]).astype(np.int64)
    rel = O.get_rel_inds_eval(im)
    if len(rel) and rng.rand() < 0.5:
        rel = rel[rng.rand(len(rel)) > rng.uniform(0.1, 0.8)]
    if len(rel) == 0:
        return
    N, E = len(im), len(rel)
    g = torch.Generator().manual_seed(int(rng.randint(1 << 30)))
    gw, gb = torch.randn(4, 2 * H, generator=g) / (H ** 0.5), torch.randn(4, generator=g)
    gwd = cu(gw)
    def state(M, dot_w):
        gi, gh, hp = torch.randn(M, 3 * H, generator=g), torch.randn(M, 3 * H, generator=g), torch.randn(M, H, generator=g)
        return ops.gru_gate(cu(gi), cu(gh), None, cu(hp.to(dtype)), dtype, dot_w=dot_w)
    v_d, nd = state(N, gwd[:, :H])
    e_d, ed = state(E, gwd[:, H:])
    v, e = v_d.float().cpu(), e_d.float().cpu()
    per = [int((rel[:, 0] == b).sum()) for b in range(nb)]
    csr = ops.edge_csr(cu(rel), N, cu(im), graphs=(nb, max(sizes), max(per)))
    s, o = torch.from_numpy(rel[:, 1]), torch.from_numpy(rel[:, 2])
    gt = [torch.sigmoid(torch.cat((a, e), 1) @ gw[k] + gb[k]) for k, a in enumerate((v[s], v[o], v[s], v[o]))]
    exp_ein = gt[0][:, None] * v[s] + gt[1][:, None] * v[o]
    exp_out = torch.zeros(N, H).index_add_(0, s, gt[2][:, None] * e)
    exp_in = torch.zeros(N, H).index_add_(0, o, gt[3][:, None] * e)
    tol = dict(atol=3e-5, rtol=1e-5) if dtype == torch.float32 else dict(atol=4e-2, rtol=2e-2)
    ctol = dict(atol=2e-4, rtol=1e-5) if dtype == torch.float32 else dict(atol=0.5, rtol=3e-2)
    outs = [('fused', ops.imp_fused(v_d, e_d, cu(rel), csr, gwd.to(dtype), cu(gb)))]
    if ops.imp_sliced_ok(csr, H, dtype):
        outs.append(('sliced', ops.imp_sliced(v_d, e_d, csr, nd, ed, cu(gb))))
        cs = torch.empty((N, H), dtype=dtype, device=dev)
        ops.imp_sliced(v_d, e_d, csr, nd, ed, cu(gb), ctx_sum=cs)
        torch.testing.assert_close(cs.float().cpu(), exp_out + exp_in, **ctol)
    for name, (e_in, ctx2) in outs:
        torch.testing.assert_close(e_in.float().cpu(), exp_ein, **tol, msg=lambda m: '%s H=%d %s sizes=%s: %s' % (name, H, dtype, sizes, m))
        torch.testing.assert_close(ctx2[0].float().cpu(), exp_out, **ctol)
        torch.testing.assert_close(ctx2[1].float().cpu(), exp_in, **ctol)
    stats['imp'] += 1


def fuzz_gemm():
    dtype = torch.bfloat16 if rng.rand() < 0.7 else torch.float32
    ke = 64 if dtype == torch.bfloat16 else 32
    M = int(rng.choice([1, 7, 64, 100, 256, 300, 512, 1000, 2048, 4100]))
    N = int(rng.choice([8, 51, 64, 152, 256, 512, 1536, 4096]))
    K1 = ke * int(rng.randint(1, 40))
    K2 = ke * int(rng.randint(1, 8)) if rng.rand() < 0.3 else 0
    act = int(rng.rand() < 0.5)
    g = torch.Generator().manual_seed(int(rng.randint(1 << 30)))
    A, W = torch.randn(M, K1, generator=g).to(dtype), (torch.randn(N, K1 + K2, generator=g) / (K1 + K2) ** 0.5).to(dtype)
    A2 = torch.randn(M, K2, generator=g).to(dtype) if K2 else None
    bias = torch.randn(N, generator=g)
    sep = K2 and rng.rand() < 0.5
    out = ops.gemm(cu(A), cu(W[:, :K1]) if sep else cu(W), cu(bias), act, out_dtype=torch.float32, A2=cu(A2) if K2 else None,
                   W2=cu(W[:, K1:]) if sep else None)
    X = torch.cat((A.float(), A2.float()), 1) if K2 else A.float()
    ref = X @ W.float().t() + bias
    if act:
        ref = ref.relu()
    torch.testing.assert_close(out.cpu(), ref, atol=2e-3 if dtype == torch.float32 else 2e-2, rtol=2e-3 if dtype == torch.float32 else 2e-2,
                               msg=lambda m: 'gemm M=%d N=%d K=%d+%d %s act=%d sep=%s: %s' % (M, N, K1, K2, dtype, act, sep, m))
    stats['gemm'] += 1


def fuzz_tn():
    Mred = 64 * int(rng.randint(1, 60))
    N, K = 128 * int(rng.randint(1, 6)), 128 * int(rng.randint(1, 9))
    g = torch.Generator().manual_seed(int(rng.randint(1 << 30)))
    A, B = torch.randn(Mred, N, generator=g).bfloat16(), torch.randn(Mred, K, generator=g).bfloat16()
    out = ops.gemm_tn(cu(A), cu(B), splits=int(rng.choice([1, 2, 3])) if Mred >= 192 else None)
    ref = A.float().t() @ B.float()
    torch.testing.assert_close(out.cpu(), ref, atol=0.05 * (Mred ** 0.5) / 8, rtol=2e-2, msg=lambda m: 'tn Mred=%d N=%d K=%d: %s' % (Mred, N, K, m))
    stats['tn'] += 1


def run(budget=60.0, seed=0):
    global rng
    rng = np.random.RandomState(seed)
    for k in stats:
        stats[k] = 0
    t0 = time.time()
    while time.time() - t0 < budget:
        fuzz_imp()
        fuzz_gemm()
        fuzz_tn()
    torch.cuda.synchronize()
    return dict(stats)


if __name__ == '__main__':
    t0 = time.time()
    out = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(os.environ.get('FUZZ_SEED', '0')))
    print('fuzz ok:', out, 'in %.0f s' % (time.time() - t0))
