"""Unordered box pairs (sgg_amd/pairing.py, csrc/edge_pairs.hip): the union box of (s, o) and (o, s) is one box, so the pooled row and
fc6's long contraction are made once per pair.  The tables, the three kernels around them, and the whole model with the pair path
on and off: the same features bit for bit, the same scores and gradients up to the order of f32 sums."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def cu(x):
    return torch.as_tensor(x).to(DEV).contiguous()


def _edges(counts, rng, drop=0.0):
    """all ordered pairs of every image, (image, subject, object) with global box indices, sorted; a share dropped"""
    rows, first = [], 0
    for b, n in enumerate(counts):
        for s in range(n):
            for o in range(n):
                if s != o and rng.rand() >= drop:
                    rows.append((b, first + s, first + o))
        first += n
    return np.array(rows, dtype=np.int64).reshape(-1, 3)


def _segs(counts):
    out, first = [], 0
    for b, n in enumerate(counts):
        out.append((b, first, first + n))
        first += n
    return out


def test_pair_tables_and_slots():
    from sgg_amd.pairing import _tables, make_pairing
    rng = np.random.RandomState(0)
    counts = [5, 2, 9, 3, 32]
    rel = _edges(counts, rng)
    pm = make_pairing(cu(rel), _segs(counts))
    U = sum(n * (n - 1) // 2 for n in counts)
    assert pm is not None and pm.U == U and pm.E == len(rel)
    pairs = pm.pairs.cpu().numpy()
    assert len(pairs) == U and (pairs[:, 0] < pairs[:, 1]).all() and len({tuple(p) for p in pairs.tolist()}) == U
    e2u, u2e = pm.e2u.cpu().numpy(), pm.u2e.cpu().numpy()
    assert int(pm.flag.item()) == 0
    # every edge sits on the slot of its unordered pair; every slot lists exactly its two edges
    np.testing.assert_array_equal(pairs[e2u], np.stack((rel[:, 1:].min(1), rel[:, 1:].max(1)), 1))
    assert (u2e >= 0).all()
    for u in range(U):
        a, b = u2e[u]
        assert a != b and e2u[a] == u and e2u[b] == u
    # the host tables are cached per box counts
    assert _tables(counts, torch.device(DEV))[3] is pm.pairs
    # sparse edge lists keep the per-edge path; so do lists on which the host cannot rule out repeated ordered pairs
    assert make_pairing(cu(_edges(counts, rng, drop=0.5)), _segs(counts)) is None
    assert make_pairing(cu(rel), _segs(counts), max_per_pair=3) is None
    # a list with some edges missing (still dense enough): slots with one edge, or none
    rel2 = _edges(counts, rng, drop=0.2)
    pm2 = make_pairing(cu(rel2), _segs(counts))
    assert pm2 is not None and int(pm2.flag.item()) == 0
    e2u2, u2e2 = pm2.e2u.cpu().numpy(), pm2.u2e.cpu().numpy()
    cnt = np.bincount(e2u2, minlength=U)
    np.testing.assert_array_equal((u2e2 >= 0).sum(1), cnt)
    # broken promises are flagged: a third edge on a pair (bit 1), an edge outside its image (bit 0)
    from sgg_amd import ops
    first, ubase, cntt, _, _ = _tables(counts, torch.device(DEV))
    _, _, flag = ops.pair_slots(cu(np.concatenate((rel, rel[:1]))), first, ubase, cntt, U)
    assert int(flag.item()) == 2
    bad = rel.copy()
    bad[3, 2] = 5                    # object in the next image
    _, _, flag = ops.pair_slots(cu(bad), first, ubase, cntt, U)
    assert int(flag.item()) & 1


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_pair_kernels(dtype):
    from sgg_amd import ops
    g = torch.Generator().manual_seed(3)
    # gathered-row addend in the GEMM epilogue (128x128 tiles and the 256x256 ping-pong kernel)
    for M, N, K, R in ((300, 192, 128, 77), (2100, 512, 512, 1050)):
        A = torch.randn(M, K, generator=g).to(dtype)
        W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype)
        b = torch.randn(N, generator=g)
        rows = torch.randn(R, N, generator=g)
        idx = torch.randint(0, R, (M,), generator=g, dtype=torch.int32)
        ref = torch.relu(A.float() @ W.float().t() + rows[idx.long()] + b)
        out = ops.gemm_addrows(cu(A), cu(W), cu(b), cu(rows), cu(idx), ops.ACT_RELU, out_dtype=torch.float32)
        tol = dict(atol=2e-4, rtol=1e-4) if dtype == torch.float32 else dict(atol=2e-2, rtol=2e-2)
        torch.testing.assert_close(out.cpu(), ref, **tol)
        out2 = ops.gemm_addrows(cu(A[:R]), cu(W), None, cu(rows), None, out_dtype=torch.float32)      # no index: row m itself
        torch.testing.assert_close(out2.cpu(), A[:R].float() @ W.float().t() + rows, **tol)
    # pair-sum transpose: ascending edge order, -1 = no edge, zero padding
    E, C, U = 333, 200, 150
    x = torch.randn(E, C, generator=g).to(dtype)
    u2e = torch.full((U, 2), -1, dtype=torch.int32)
    perm = torch.randperm(E, generator=g)[:2 * U - 20].to(torch.int32)
    u2e.view(-1)[:len(perm)] = perm
    u2e = u2e[torch.randperm(U, generator=g)]
    out = ops.transpose_pairsum(cu(x), cu(u2e))
    assert out.shape == (C, 192) and out.dtype == dtype
    ref = torch.zeros(U, C)
    for u in range(U):
        es = sorted(int(e) for e in u2e[u] if e >= 0)
        acc = torch.zeros(C)
        for e in es:
            acc = acc + x[e].float()
        ref[u] = acc
    torch.testing.assert_close(out[:, :U].float().cpu(), ref.to(dtype).float().t(), atol=0, rtol=0)
    assert float(out[:, U:].float().abs().max()) == 0.0
    # broadcast add over column groups
    M, Cg, grp = 37, 24, 49
    y = torch.randn(M, Cg * grp, generator=g).to(dtype)
    r = torch.randn(M, Cg, generator=g)
    got = ops.group_bcast_add_(cu(y).clone(), cu(r), grp)
    ref = (y.float() + r.repeat_interleave(grp, 1)).to(dtype)
    torch.testing.assert_close(got.cpu().float(), ref.float(), atol=0, rtol=0)
    got = ops.group_bcast_add_(cu(y[:, :100]).clone(), cu(r), grp, col0=30)     # a column slice: groups counted from column 30
    ref = (y[:, :100].float() + r.repeat_interleave(grp, 1)[:, 30:130]).to(dtype)
    torch.testing.assert_close(got.cpu().float(), ref.float(), atol=0, rtol=0)


def test_weight_gradient_gemm_with_group_addend_in_its_epilogue():
    """gemm_full_waves(gadd=...): whole rounds of the ping-pong kernel with the addend in the epilogue + the split-K tail columns with
    the addend applied afterwards = GEMM + broadcast add"""
    from sgg_amd import ops
    g = torch.Generator().manual_seed(8)
    M, K, grp = 512, 2048, 49
    for N in (66048, 1024):                 # 516 tiles: 2 rounds + 4 tiles of tail; 8 tiles: one launch
        A = (torch.randn(M, K, generator=g) / 8).to(torch.bfloat16)
        W = (torch.randn(N, K, generator=g) / 8).to(torch.bfloat16)
        C = (N + grp - 1) // grp
        r = torch.randn(M, C, generator=g)
        out = ops.gemm_full_waves(cu(A), cu(W), out_dtype=torch.float32, gadd=(cu(r), grp))
        plain = ops.gemm_full_waves(cu(A), cu(W), out_dtype=torch.float32)
        ref = plain.cpu() + r.repeat_interleave(grp, 1)[:, :N]
        torch.testing.assert_close(out.cpu(), ref, atol=1e-4, rtol=1e-5)
        n4 = min(N, 4096)
        chk = A[:64].float() @ W[:n4].float().t() + r[:64].repeat_interleave(grp, 1)[:, :n4]
        torch.testing.assert_close(out[:64, :n4].cpu(), chk, atol=2e-2, rtol=2e-3)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_model_with_and_without_pairs(dtype, monkeypatch):
    """eval scores and every training gradient with the pair path against the per-edge path (SGG_EDGE_PAIRS=0); the pooled rows are
    the per-edge kernel's rows bit for bit"""
    import sgg_amd
    from sgg_amd.pairing import PairedEdgeFeat
    from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
    from sgg_amd.train import param_names
    S = 96
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls', min_size=S, max_size=S)).to(DEV)
    model.set_compute_dtype(dtype)
    batch = synthetic_batch(B=3, S=S, n_boxes=7, n_fg=3, seed=4, ragged=True)
    out = {}
    for flag in ('1', '0'):
        monkeypatch.setenv('SGG_EDGE_PAIRS', flag)
        model.train()
        model.dropout_p = 0.0
        sd_bn = {k: v.clone() for k, v in model.state_dict().items() if 'running' in k or 'tracked' in k}
        res = model([tuple(batch)])
        assert isinstance(res.edge_feat, PairedEdgeFeat) == (flag == '1')
        g = torch.Generator().manual_seed(0)
        Wo = torch.randn(res.rm_obj_dists.shape, generator=g).to(DEV)
        Wr = torch.randn(res.rel_dists.shape, generator=g).to(DEV)
        model.zero_grad()
        ((res.rm_obj_dists * Wo).sum() + (res.rel_dists * Wr).sum()).backward()
        named = dict(model.named_parameters())
        out[flag] = (res.rm_obj_dists.detach().clone(), res.rel_dists.detach().clone(),
                     {n: named[n].grad.detach().clone() for n in param_names(model)}, res.edge_feat.float().clone())
        model.load_state_dict(sd_bn, strict=False)      # the same BatchNorm running statistics for the second pass
    assert torch.equal(out['1'][3], out['0'][3])                     # PairedEdgeFeat.dense() = the reference's per-edge tensor
    tol = dict(atol=2e-4, rtol=2e-4) if dtype == torch.float32 else dict(atol=6e-2, rtol=6e-2)
    torch.testing.assert_close(out['1'][0], out['0'][0], **tol)
    torch.testing.assert_close(out['1'][1], out['0'][1], **tol)
    for n in out['1'][2]:
        a, b = out['1'][2][n].float(), out['0'][2][n].float()
        scale = float(b.abs().max()) + 1e-6
        err = float((a - b).abs().max()) / scale
        assert err < (2e-4 if dtype == torch.float32 else 4e-2), (n, err, scale)
    # eval forward: same detections
    model.eval()
    ev = {}
    for flag in ('1', '0'):
        monkeypatch.setenv('SGG_EDGE_PAIRS', flag)
        ev[flag] = model([tuple(b_ for b_ in batch)])
    for a, b in zip(ev['1'], ev['0']):
        if dtype == torch.float32:
            np.testing.assert_allclose(a, b, atol=2e-4, rtol=2e-4)
        else:
            assert a.shape == b.shape
