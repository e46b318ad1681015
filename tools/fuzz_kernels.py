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
stats = {'imp': 0, 'gemm': 0, 'tn': 0, 'tn256': 0, 'conv': 0, 'roi': 0, 'raster': 0, 'pairs': 0, 'unordered': 0}


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
    ctol = dict(atol=2e-4, rtol=1e-5) if dtype == torch.float32 else dict(atol=0.5, rtol=3e-2) if dtype == torch.bfloat16 else dict(atol=6e-2, rtol=4e-3)
    plain = ops.edge_csr(cu(rel), N, cu(im))
    for pair in (2, 0):                               # the forward's context sums / the backward's projection gradient
        exp_out = torch.zeros(N, H).index_add_(0, s, gt[pair][:, None] * e)
        exp_in = torch.zeros(N, H).index_add_(0, o, gt[pair + 1][:, None] * e)
        outs = [('lists', ops.imp_ctx(e_d, plain, N, nd, ed, cu(gb), pair=pair))]
        if ops.imp_sliced_ok(csr, H, dtype):
            outs.append(('promised', ops.imp_ctx(e_d, csr, N, nd, ed, cu(gb), pair=pair)))
            cs = torch.empty((N, H), dtype=dtype, device=dev)
            ops.imp_ctx(e_d, csr, N, nd, ed, cu(gb), pair=pair, ctx_sum=cs)
            torch.testing.assert_close(cs.float().cpu(), exp_out + exp_in, **ctol)
        for name, ctx2 in outs:
            msg = lambda m: '%s H=%d %s sizes=%s pair=%d: %s' % (name, H, dtype, sizes, pair, m)  # noqa: E731
            torch.testing.assert_close(ctx2[0].float().cpu(), exp_out, msg=msg, **ctol)
            torch.testing.assert_close(ctx2[1].float().cpu(), exp_in, msg=msg, **ctol)
    # the edge GRU from the node projection against the gate kernel on materialised pre-activations
    P, gh, b_ih = torch.randn(N, 3 * H, generator=g), torch.randn(E, 3 * H, generator=g), torch.randn(3 * H, generator=g) / 4
    gi = gt[0][:, None] * P[s] + gt[1][:, None] * P[o] + b_ih
    got = ops.gru_gate_proj(cu(gh), cu(P), cu(b_ih), csr, nd, ed, cu(gb), e_d)
    ref = ops.gru_gate(cu(gi), cu(gh), None, e_d, dtype)
    torch.testing.assert_close(got.float(), ref.float(), **(dict(atol=1e-5, rtol=1e-5) if dtype == torch.float32 else dict(atol=3e-2, rtol=2e-2)))
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


def fuzz_tn256():
    """sgg_gemm_tn256 (the ping-pong kernel's TN form): whole 256 x 256 tiles, ANY number of reduction rows (the zero-padded tail tile),
    both 16-bit types, optional group addend, column blocks of wider operands, 16-bit / fp32 output"""
    Mred = int(rng.randint(1, 1500))
    N, K = 256 * int(rng.randint(1, 4)), 256 * int(rng.randint(1, 5))
    dtype = torch.float16 if rng.rand() < 0.5 else torch.bfloat16
    g = torch.Generator().manual_seed(int(rng.randint(1 << 30)))
    extra = 256 * int(rng.randint(0, 2))
    A = (torch.randn(Mred, N, generator=g) / 4).to(dtype)
    Bw = torch.randn(Mred, K + extra, generator=g).to(dtype)
    B = Bw[:, extra:]
    ref = cu(A).float().t() @ cu(Bw)[:, extra:].float()
    gadd = None
    if rng.rand() < 0.5:
        group = int(rng.choice([8, 49, 64]))
        r = torch.randn(N, (K + extra) // group + 2, generator=g)
        gadd = (cu(r), group)
        ref = ref + cu(r)[:, (torch.arange(K, device=dev) + extra) // group]
    out_dtype = torch.float32 if rng.rand() < 0.5 else dtype
    out = ops.gemm_tn256(cu(A), cu(Bw)[:, extra:], out_dtype=out_dtype, gadd=gadd, col0=extra if gadd is not None else 0)
    tol = dict(atol=2e-3 * (Mred ** 0.5) + 1e-3, rtol=1e-4) if out_dtype == torch.float32 else dict(atol=0.03 * (Mred ** 0.5) + 0.05, rtol=2e-2)
    torch.testing.assert_close(out.float(), ref, msg=lambda m: 'tn256 Mred=%d N=%d K=%d %s gadd=%s: %s' % (Mred, N, K, dtype, gadd is not None, m), **tol)
    stats['tn256'] = stats.get('tn256', 0) + 1


def fuzz_conv():
    """3x3 conv + ReLU (+ fused 2x2 max-pool): spatial LDS-patch kernel (maps >= 64) and implicit GEMM (small maps)"""
    dtype = torch.bfloat16 if rng.rand() < 0.7 else torch.float32
    B = int(rng.randint(1, 3))
    H, W = int(rng.randint(6, 90)), int(rng.randint(6, 90))
    if rng.rand() < 0.4:
        H, W = H & ~1, W & ~1
    Cin, Cout = int(rng.choice([64, 128])), int(rng.choice([64, 128, 256]))
    g = torch.Generator().manual_seed(int(rng.randint(1 << 30)))
    x = torch.randn(B, Cin, H, W, generator=g).to(dtype)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(dtype)
    b = torch.randn(Cout, generator=g)
    ref = torch.nn.functional.conv2d(x.float(), w.float(), b, padding=1).relu()
    xp = torch.zeros(B, H + 2, W + 2, Cin, dtype=dtype)
    xp[:, 1:-1, 1:-1] = x.permute(0, 2, 3, 1)
    tol = 2e-4 if dtype == torch.float32 else 4e-2
    out = torch.zeros((B, H + 2, W + 2, Cout), dtype=dtype, device=dev)
    ops.conv3x3_relu(cu(xp), cu(w.permute(0, 2, 3, 1)), cu(b), out, 1)
    torch.testing.assert_close(out[:, 1:-1, 1:-1].float().cpu().permute(0, 3, 1, 2), ref, atol=tol, rtol=tol,
                               msg=lambda m: 'conv B=%d %dx%d %d->%d %s: %s' % (B, H, W, Cin, Cout, dtype, m))
    if H % 2 == 0 and W % 2 == 0 and ops.conv_pool_fusable(H, W, Cout):
        got = torch.zeros((B, H // 2 + 2, W // 2 + 2, Cout), dtype=dtype, device=dev)
        ops.conv3x3_relu(cu(xp), cu(w.permute(0, 2, 3, 1)), cu(b), got, 1, pool=True)
        refp = torch.nn.functional.max_pool2d(out[:, 1:-1, 1:-1].float().permute(0, 3, 1, 2), 2)
        # (the unpooled map may come from the implicit-GEMM kernel -- another summation order than the patch kernel's: one bf16 ulp)
        exact = Cout < 256 or dtype == torch.float32 and False
        torch.testing.assert_close(got[:, 1:-1, 1:-1].float().permute(0, 3, 1, 2), refp, atol=0 if exact else tol, rtol=0 if exact else tol)
    stats['conv'] += 1


def rand_boxes(n, S):
    xy = rng.uniform(0, 0.7 * S, size=(n, 2))
    wh = rng.uniform(1, 0.5 * S, size=(n, 2))
    bx = np.concatenate((xy, np.minimum(xy + wh, S - 1)), 1).astype(np.float32)
    if n > 2 and rng.rand() < 0.3:
        bx[0, 2:] = bx[0, :2]                       # degenerate box
        bx[1] = [0, 0, S - 1, S - 1]                # whole image
    return bx


def fuzz_roi():
    dtype = torch.bfloat16 if rng.rand() < 0.5 else torch.float32
    B, C = int(rng.randint(1, 4)), int(rng.choice([64, 128]))
    Hf, Wf = int(rng.randint(5, 40)), int(rng.randint(5, 40))
    N = int(rng.randint(2, 14))
    fm = torch.from_numpy(rng.randn(B, C, Hf, Wf).astype(np.float32)).to(dtype)
    boxes = rand_boxes(N, 16 * min(Hf, Wf))
    im = np.sort(rng.randint(0, B, N)).astype(np.float32)
    rois = np.concatenate((im[:, None], boxes), 1)
    pairs = np.array([(i, j) for i in range(N) for j in range(N) if i != j and im[i] == im[j]], np.int64).reshape(-1, 2)
    exp_n, exp_e = O.node_edge_features(fm.float().numpy(), rois, pairs if len(pairs) else np.zeros((0, 2), np.int64))
    fmd = cu(fm.permute(0, 2, 3, 1).contiguous())
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    got = ops.roi_align(fmd, cu(rois))
    np.testing.assert_allclose(got.float().cpu().numpy().reshape(exp_n.shape), exp_n, atol=tol, rtol=tol)
    if len(pairs):
        gote = ops.roi_align(fmd, cu(rois), cu(pairs))
        np.testing.assert_allclose(gote.float().cpu().numpy().reshape(exp_e.shape), exp_e, atol=tol, rtol=tol)
    stats['roi'] += 1


def fuzz_raster():
    N = int(rng.randint(2, 12))
    B = int(rng.randint(1, 3))
    S = float(rng.choice([100, 592, 1024]))
    boxes = rand_boxes(N, S)
    boxes[:, 2:] = np.maximum(boxes[:, 2:], boxes[:, :2] + 1)     # positive extent (a zero extent divides by zero in both)
    im = np.sort(rng.randint(0, B, N)).astype(np.float32)
    rois = np.concatenate((im[:, None], boxes), 1).astype(np.float32)
    pairs = np.array([(i, j) for i in range(N) for j in range(N) if i != j and im[i] == im[j]], np.int64).reshape(-1, 2)
    if not len(pairs):
        return
    pr = np.concatenate((rois[pairs[:, 0], 1:], rois[pairs[:, 1], 1:]), 1)
    np.testing.assert_array_equal(ops.union_rects(cu(rois), cu(pairs), 27, 0.0).cpu().numpy(), O.draw_union_boxes(pr, 27))
    sizes = [(int(S) + 10 * b, int(S) + 7) for b in range(B)]
    np.testing.assert_allclose(ops.union_rects(cu(rois), cu(pairs), 27, 0.0, im_sizes=sizes).cpu().numpy(),
                               O.draw_union_boxes_grid(rois, pairs, sizes, 27), atol=2e-5)
    stats['raster'] += 1


def fuzz_pairs():
    B = int(rng.randint(1, 6))
    sizes = [int(rng.randint(1, 12)) for _ in range(B)]
    im = np.concatenate([np.full(n, b) for b, n in enumerate(sizes)]).astype(np.int64)
    boxes = rand_boxes(len(im), 200)
    for ov in (False, True):
        exp = O.get_rel_inds_eval(im, boxes, ov)
        out, cnt = ops.pair_index_eval(cu(im), cu(boxes) if ov else None, ov)
        n = int(cnt.item())
        if len(exp) and not (exp.shape == (1, 3) and n == 0):
            np.testing.assert_array_equal(out[:n].cpu().numpy(), exp)
    stats['pairs'] += 1


def fuzz_unordered():
    """the unordered-pair tables (closed form on complete / thinned lists, symmetric sparse lists) and the kernels around them"""
    from sgg_amd.pairing import make_pairing, make_pairing_symmetric
    counts = [int(rng.randint(1, 14)) for _ in range(int(rng.randint(1, 6)))]
    rows, segs, first = [], [], 0
    sym = rng.rand() < 0.5
    dropm = None
    for b, n in enumerate(counts):
        keep = np.triu(rng.rand(n, n) < rng.uniform(0.3, 1.0), 1)
        keep = keep | keep.T
        for s_ in range(n):
            for o_ in range(n):
                if s_ != o_ and (keep[s_, o_] if sym else rng.rand() < 0.9):
                    rows.append((b, first + s_, first + o_))
        segs.append((b, first, first + n))
        first += n
    if len(rows) < 4:
        return
    rel = np.array(rows, dtype=np.int64)
    pm = make_pairing_symmetric(cu(rel), first, min_edges=2) if sym else make_pairing(cu(rel), segs)
    if pm is None:
        return
    assert int(pm.flag.item()) == 0
    pairs, e2u, u2e = pm.pairs.cpu().numpy(), pm.e2u.cpu().numpy(), pm.u2e.cpu().numpy()
    np.testing.assert_array_equal(pairs[e2u], np.stack((rel[:, 1:].min(1), rel[:, 1:].max(1)), 1))
    cnt = np.bincount(e2u, minlength=pm.U)
    np.testing.assert_array_equal((u2e >= 0).sum(1), cnt)
    for u in range(pm.U):
        for e_ in u2e[u]:
            assert e_ < 0 or e2u[e_] == u
    # pair-sum transpose + gathered-row GEMM on these tables
    dtype = torch.bfloat16 if rng.rand() < 0.6 else torch.float32
    g = torch.Generator().manual_seed(int(rng.randint(1 << 30)))
    C = int(rng.choice([64, 192, 256]))
    x = torch.randn(len(rel), C, generator=g).to(dtype)
    out = ops.transpose_pairsum(cu(x), pm.u2e)
    ref = torch.zeros(pm.U, C)
    ref.index_add_(0, torch.from_numpy(e2u).long(), x.float())
    torch.testing.assert_close(out[:, :pm.U].float().cpu().t(), ref.to(dtype).float(), atol=2e-2 if dtype == torch.bfloat16 else 1e-5, rtol=1e-2)
    K, N = 64 * int(rng.randint(1, 5)), 64 * int(rng.randint(1, 9))
    A = torch.randn(len(rel), K, generator=g).to(dtype)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype)
    b = torch.randn(N, generator=g)
    yu = torch.randn(pm.U, N, generator=g)
    got = ops.gemm_addrows(cu(A), cu(W), cu(b), cu(yu), pm.e2u, ops.ACT_RELU, out_dtype=torch.float32)
    exp = torch.relu(A.float() @ W.float().t() + yu[torch.from_numpy(e2u).long()] + b)
    torch.testing.assert_close(got.cpu(), exp, atol=3e-2 if dtype == torch.bfloat16 else 3e-4, rtol=2e-2 if dtype == torch.bfloat16 else 1e-4)
    stats['unordered'] += 1


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
        fuzz_tn256()
        fuzz_conv()
        fuzz_roi()
        fuzz_raster()
        fuzz_pairs()
        fuzz_unordered()
    torch.cuda.synchronize()
    return dict(stats)


if __name__ == '__main__':
    t0 = time.time()
    out = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(os.environ.get('FUZZ_SEED', '0')))
    print('fuzz ok:', out, 'in %.0f s' % (time.time() - t0))
