"""The x3 mode (sgg_amd/ops.py "the x3 mode", sgg_split3): fp32 storage, every MFMA contraction on f16 SPLIT operands
(x = hi + lo; hi.hi + hi.lo + lo.hi accumulated in fp32 by one f16 GEMM over 3 K columns).  Results must be fp32-grade: contractions
within ~1e-6 relative of float64, the model's logits within the north star's 1e-3 of the CPU oracle (in fact within 1e-4 of the
exact-fp32 mode), gradients equal to the exact mode's, a Trainer run."""
import numpy as np
import pytest
import torch

from oracle import sgg_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
S = 96


@pytest.fixture()
def ops():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from sgg_amd import ops
    prev = ops.set_split3(False)
    yield ops
    ops.set_split3(prev)


def test_split3_halves_reconstruct_the_operand(ops):
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(37, 200, generator=g) * torch.logspace(-4, 2, 200)).to(DEV)       # magnitudes over six decades
    for weights in (False, True):
        s3 = ops.split3(x, weights=weights).float()
        Kp = 256          # K rounded up to 64
        assert s3.shape == (37, 3 * Kp)
        hi, a, b = s3[:, :200], s3[:, Kp:Kp + 200], s3[:, 2 * Kp:2 * Kp + 200]
        lo = a if weights else b
        assert torch.equal(hi, b if weights else a)
        assert float(s3[:, 200:Kp].abs().max()) == 0 and float(s3[:, Kp + 200:2 * Kp].abs().max()) == 0
        assert torch.equal(hi, x.half().float())
        err = (hi.double() + lo.double() - x.double()).abs()
        # 22 significand bits where lo is a normal f16 number; an absolute floor of half an f16 subnormal step (3e-8) below that
        assert bool((err <= x.double().abs() * 2.0 ** -21 + 3.1e-8).all()), float(err.max())


@pytest.mark.parametrize('M,N,K', [(300, 520, 512), (1024, 4096, 25088), (256, 151, 4096), (3968, 256, 98)])
def test_x3_gemm_is_fp32_grade(ops, M, N, K):
    g = torch.Generator().manual_seed(M + K)
    Kp = (K + 31) // 32 * 32          # (the exact-fp32 MFMA kernel's K-tile: 128 bytes)
    A = torch.zeros(M, Kp)
    W = torch.zeros(N, Kp)
    A[:, :K] = torch.randn(M, K, generator=g).relu() * 3
    W[:, :K] = torch.randn(N, K, generator=g) * (2.0 / K) ** 0.5
    bias = torch.randn(N, generator=g)
    ref = (A.double() @ W.double().t() + bias.double()).relu()
    A, W, bias = A.to(DEV), W.to(DEV), bias.to(DEV)
    exact = ops.gemm(A, W, bias=bias, act=ops.ACT_RELU).double().cpu()
    ops.set_split3(True)
    got = ops.gemm(A, W, bias=bias, act=ops.ACT_RELU)
    ops.set_split3(False)
    assert got.dtype == torch.float32
    scale = float(ref.abs().max())
    e_x3, e_f32 = float((got.double().cpu() - ref).abs().max()) / scale, float((exact - ref).abs().max()) / scale
    print('K=%d: x3 %.2e, exact-fp32 MFMA %.2e (relative to the largest output)' % (K, e_x3, e_f32))
    assert e_x3 <= max(4e-6, 2.0 * e_f32), (e_x3, e_f32)      # fp32-grade: no worse than twice the exact-fp32 MFMA's own summation error
    # two K segments (fc6's fold) and the gathered-row addend of the pair path
    A2 = (torch.randn(M, 64, generator=g)).to(DEV)
    W2 = (torch.randn(N, 64, generator=g) / 8).to(DEV)
    ref2 = A.double().cpu() @ W.double().cpu().t() + A2.double().cpu() @ W2.double().cpu().t()
    ops.set_split3(True)
    got2 = ops.gemm(A, W, A2=A2, W2=W2)
    rows = torch.randn(7, N, generator=g).to(DEV)
    idx = torch.randint(0, 7, (M,), generator=g).int().to(DEV)
    got3 = ops.gemm_addrows(A, W, bias, rows, idx)
    ops.set_split3(False)
    assert float((got2.double().cpu() - ref2).abs().max()) <= max(4e-6, 2.0 * e_f32) * float(ref2.abs().max())
    ref3 = A.double().cpu() @ W.double().cpu().t() + rows.double().cpu()[idx.long().cpu()] + bias.double().cpu()
    assert float((got3.double().cpu() - ref3).abs().max()) <= max(4e-6, 2.0 * e_f32) * float(ref3.abs().max())


def test_x3_conv_is_fp32_grade(ops):
    B, H, Wd, Ci, Co = 2, 76, 76, 64, 128
    g = torch.Generator().manual_seed(5)
    x = torch.zeros(B, H + 2, Wd + 2, Ci)
    x[:, 1:-1, 1:-1] = torch.randn(B, H, Wd, Ci, generator=g).relu()
    w = torch.randn(Co, 3, 3, Ci, generator=g) * (2.0 / (9 * Ci)) ** 0.5
    b = torch.randn(Co, generator=g) * 0.1
    ref = torch.nn.functional.conv2d(x[:, 1:-1, 1:-1].permute(0, 3, 1, 2).double(), w.permute(0, 3, 1, 2).double(), b.double(), padding=1).relu()
    ops.set_split3(True)
    out = torch.zeros(B, H + 2, Wd + 2, Co, device=DEV)
    ops.conv3x3_relu(x.to(DEV), w.to(DEV), b.to(DEV), out, 1)
    ops.set_split3(False)
    got = out[:, 1:-1, 1:-1].permute(0, 3, 1, 2).double().cpu()
    assert float(out[:, 0].abs().max()) == 0 and float(out[:, :, -1].abs().max()) == 0          # the zero border is untouched
    assert float((got - ref).abs().max()) <= 4e-6 * float(ref.abs().max())


def test_x3_model_forward_gradients_and_trainer(ops):
    import sgg_amd
    from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
    from sgg_amd.trainer import Trainer
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls', min_size=S, max_size=S))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(DEV).eval()
    batch = synthetic_batch(B=3, S=S, n_boxes=7, n_fg=3, seed=5, ragged=True)
    with torch.no_grad():
        ref = O.forward_gtbox(batch[0], batch[3], batch[4], batch[5], sd, min_size=S, max_size=S)['dets']
        model.set_compute_dtype(torch.float32)
        exact = model([tuple(batch)])
        model.set_compute_dtype(torch.float32, split3=True)
        assert ops.split3_on() and model.split3
        got = model([tuple(batch)])
    np.testing.assert_array_equal(got[3], ref[3])
    np.testing.assert_allclose(got[4], ref[4], atol=1e-3)                 # the north star's bar, against the CPU oracle
    np.testing.assert_allclose(got[2], ref[2], atol=1e-3)
    np.testing.assert_allclose(got[4], exact[4], atol=1e-4)               # and as good as the exact-fp32 MFMA mode
    # gradients: train-mode forward + backward in both modes
    model.train()
    model.dropout_p = 0.0
    grads = {}
    for split in (False, True):
        model.load_state_dict(sd)
        model.set_compute_dtype(torch.float32, split3=split)
        res = model([tuple(batch)])
        g = torch.Generator().manual_seed(0)
        Wo = torch.randn(res.rm_obj_dists.shape, generator=g).to(DEV)
        Wr = torch.randn(res.rel_dists.shape, generator=g).to(DEV)
        model.zero_grad()
        ((res.rm_obj_dists * Wo).sum() + (res.rel_dists * Wr).sum()).backward()
        grads[split] = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    assert len(grads[True]) == len(grads[False]) >= 30
    for n, ge in grads[False].items():
        gx = grads[True][n]
        assert float((gx - ge).abs().max()) <= 2e-4 * float(ge.abs().max()) + 1e-7, n
    # a Trainer run in the x3 mode
    model.load_state_dict(sd)
    model.set_compute_dtype(torch.float32, split3=True)
    tr = Trainer(model, lr=2e-2)
    losses = [float(tr.step(tuple(batch))) for _ in range(8)]
    tr.flush()
    assert losses[-1] < 0.9 * losses[0], losses
    assert all(bool(torch.isfinite(p).all()) for p in model.parameters())
    model.set_compute_dtype(torch.float16)
    model.eval()
    model.dropout_p = 0.5
    assert not ops.split3_on()


def test_pair_operand_gemm_conv_pool_kernels(ops):
    """Round 6: the x3 contractions on PAIR operands (SGG_PAIR16: rows / pixels as two f16 planes [hi | lo]; the kernels walk hi.hi + hi.lo
    + lo.hi themselves).  sgg_split2 against its definition; the pair GEMM (tile kernel, ping-pong kernel, split-K; K not a multiple of 64;
    bias + ReLU + per-channel affine) against a float64 product at the x3 mode's accuracy and against round 5's [hi | hi | lo] form; the
    pair convolution (pair plane in, pair plane or fp32 out) against torch's fp32 convolution; the pair max pool; conv1_1's pair output."""
    import torch.nn.functional as F
    from sgg_amd import _lib
    g = torch.Generator().manual_seed(7)
    # ---- split2
    x = (torch.randn(37, 200, generator=g) * torch.logspace(-3, 2, 200)).to(DEV)
    p2 = ops.split2(x)
    assert p2.shape == (37, 2 * 256) and p2.dtype == torch.float16
    hi = x.half()
    assert torch.equal(p2[:, :200], hi) and torch.equal(p2[:, 256:456], (x - hi.float()).half())
    assert float(p2[:, 200:256].abs().max()) == 0 and float(p2[:, 456:].abs().max()) == 0
    assert float((p2[:, :200].float() + p2[:, 256:456].float() - x).abs().max()) <= 2e-6 * float(x.abs().max())
    # ---- GEMM
    ops.set_split3(True)
    try:
        for M, N, K, bias, act, affine in ((300, 200, 1000, True, ops.ACT_RELU, True),        # 128x128 tile kernel, K padded to 1024
                                           (2048, 4096, 512, True, ops.ACT_NONE, False),     # 256x256 ping-pong kernel
                                           (256, 4096, 4096, True, ops.ACT_RELU, False),     # split-K (short M, long K)
                                           (64, 40, 64, False, ops.ACT_NONE, False)):
            A = torch.randn(M, K, generator=g).to(DEV)
            W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
            b = torch.randn(N, generator=g).to(DEV) if bias else None
            ps = (torch.rand(N, generator=g) + 0.5).to(DEV) if affine else None
            pt = torch.randn(N, generator=g).to(DEV) if affine else None
            want = A.double() @ W.double().t()
            if b is not None:
                want = want + b.double()
            if act == ops.ACT_RELU:
                want = want.clamp_min(0)
            if affine:
                want = want * ps.double() + pt.double()
            ops.PAIR_GEMM = True
            got = ops.gemm(A, W, b, act, post_scale=ps, post_shift=pt)
            ops.PAIR_GEMM = False
            old = ops.gemm(A, W, b, act, post_scale=ps, post_shift=pt)
            ops.PAIR_GEMM = True
            scale = float(want.abs().max())
            assert got.dtype == torch.float32 and float((got.double() - want).abs().max()) <= 2e-5 * scale, (M, N, K, float((got.double() - want).abs().max()), scale)
            assert float((got - old).abs().max()) <= 4e-6 * scale, (M, N, K)          # same three products, another order
    finally:
        ops.PAIR_GEMM = True
        ops.set_split3(False)
    # ---- convolution on pair planes: [B, H+2, W+2, 2 Cin] -> pair plane / fp32
    for B, H, W, Cin, Cout in ((2, 38, 38, 128, 256), (1, 80, 72, 64, 64), (8, 76, 76, 256, 512)):
        x = torch.randn(B, Cin, H, W, generator=g).relu().to(DEV)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(DEV)
        b = torch.randn(Cout, generator=g).to(DEV)
        want = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1)).permute(0, 2, 3, 1)       # NHWC
        xp = torch.zeros(B, H + 2, W + 2, 2 * Cin, dtype=torch.float16, device=DEV)
        xn = x.permute(0, 2, 3, 1).contiguous()
        xp[:, 1:-1, 1:-1] = ops.split2(xn.reshape(-1, Cin)).view(B, H, W, 2 * Cin)
        wk = w.permute(0, 2, 3, 1).contiguous()                                                          # [Cout, ky, kx, Cin]
        out32 = torch.empty(B, H, W, Cout, dtype=torch.float32, device=DEV)
        ops.conv3x3_relu_pair(xp, wk, b, out32, 0, pair_out=False)
        scale = float(want.abs().max())
        assert float((out32.double() - want).abs().max()) <= 2e-5 * scale, (B, H, Cin, Cout, float((out32.double() - want).abs().max()), scale)
        outp = torch.zeros(B, H + 2, W + 2, 2 * Cout, dtype=torch.float16, device=DEV)
        ops.conv3x3_relu_pair(xp, wk, b, outp, 1, pair_out=True)
        inner = outp[:, 1:-1, 1:-1]
        assert torch.equal(inner[..., :Cout], out32.half())                                              # hi = f16(value), lo = f16(value - hi)
        assert torch.equal(inner[..., Cout:], (out32 - out32.half().float()).half())
        assert float(outp[:, 0].abs().max()) == 0 and float(outp[:, :, -1].abs().max()) == 0             # the zero border is left alone
        if H % 2 == 0 and W % 2 == 0:
            pooled = torch.zeros(B, H // 2 + 2, W // 2 + 2, 2 * Cout, dtype=torch.float16, device=DEV)
            ops.maxpool2x2_pair(outp, pooled, 1)
            val = inner[..., :Cout].float() + inner[..., Cout:].float()
            wantp = F.max_pool2d(val.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
            gotp = pooled[:, 1:-1, 1:-1]
            assert torch.equal(gotp[..., :Cout].float() + gotp[..., Cout:].float(), wantp)
    # ---- conv1_1 with a pair plane as output
    B, H, W = 2, 48, 64
    img = torch.zeros(B, H + 2, W + 2, 4, device=DEV)
    img[:, 1:-1, 1:-1, :3] = torch.randn(B, H, W, 3, generator=g).to(DEV)
    w1 = (torch.randn(64, 27, generator=g) / 5).to(DEV)
    b1 = torch.randn(64, generator=g).to(DEV)
    ref = torch.zeros(B, H + 2, W + 2, 64, device=DEV)
    ops.conv1_1(img, w1, b1, ref)
    pp = torch.zeros(B, H + 2, W + 2, 128, dtype=torch.float16, device=DEV)
    ops.conv1_1(img, w1, b1, pp, pair=True)
    assert torch.equal(pp[..., :64], ref.half()) and torch.equal(pp[..., 64:], (ref - ref.half().float()).half())


@pytest.mark.parametrize('shape', [(2, 152, 152, 128, 256, False), (2, 152, 152, 256, 256, True), (1, 304, 304, 64, 128, False),
                                   (1, 304, 296, 128, 128, True), (3, 76, 76, 256, 512, True), (1, 100, 70, 64, 256, False), (2, 304, 304, 64, 64, True), (1, 96, 120, 64, 64, False), (1, 80, 80, 128, 192, True)])
def test_x3_convolution_on_the_patch_kernel(ops, shape):
    """sgg_conv3x3_relu_x3 (conv_pp.hip, X3 form: pair plane in, weights [hi | lo | hi] per tap, pair plane out from the fp32 accumulator,
    optional fused 2x2 max pool) against a float64 convolution and against the implicit-GEMM pair form (same products, another order)"""
    import torch.nn.functional as F
    B, H, W, Cin, Cout, pool = shape
    g = torch.Generator().manual_seed(H + Cin)
    x = torch.randn(B, Cin, H, W, generator=g).relu().to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    want = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    if pool:
        want = F.max_pool2d(want, 2)
    want = want.permute(0, 2, 3, 1)
    xp = torch.zeros(B, H + 2, W + 2, 2 * Cin, dtype=torch.float16, device=DEV)
    xp[:, 1:-1, 1:-1] = ops.split2(x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous()).view(B, H, W, 2 * Cin)
    wk = w.permute(0, 2, 3, 1).contiguous()
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    out = torch.zeros(B, Ho + 2, Wo + 2, 2 * Cout, dtype=torch.float16, device=DEV)
    assert ops.conv_pp_x3_ok(H, W, Cin, Cout)
    ops.conv3x3_relu_x3pp(xp, wk, b, out, 1, pool=pool)
    inner = out[:, 1:-1, 1:-1]
    got = inner[..., :Cout].double() + inner[..., Cout:].double()
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2e-5 * scale, (shape, float((got - want).abs().max()), scale)
    assert float(out[:, 0].abs().max()) == 0 and float(out[:, -1].abs().max()) == 0 and float(out[:, :, 0].abs().max()) == 0 and float(out[:, :, -1].abs().max()) == 0
    ref = torch.zeros(B, H + 2, W + 2, 2 * Cout, dtype=torch.float16, device=DEV)
    ops.conv3x3_relu_pair(xp, wk, b, ref, 1, pair_out=True)
    if pool:
        pooled = torch.zeros_like(out)
        ops.maxpool2x2_pair(ref, pooled, 1)
        ref = pooled
    r = ref[:, 1:-1, 1:-1]
    assert float((got - (r[..., :Cout].double() + r[..., Cout:].double())).abs().max()) <= 4e-6 * scale


@pytest.mark.parametrize('shape', [(3968, 4096, 25088), (7936, 512, 4096), (256, 512, 1536), (96, 256, 256)])
def test_x3_weight_gradient_contraction_on_pair_operands_as_they_lie(ops, shape):
    """ops.gemm_tn_x3: dW = dY^T X for fp32 dY [M, a], X [M, b] in the x3 arithmetic through the ping-pong kernel's TN form on PAIR operands (no
    transposed copies): against a float64 product, and against the transposes + NT form it replaces"""
    M, a, b = shape
    g = torch.Generator().manual_seed(M + a)
    dY = (torch.randn(M, a, generator=g) / 30).to(DEV)
    X = torch.randn(M, b, generator=g).relu().to(DEV)
    assert ops.gemm_tn_x3_ok(dY, X) == ((a // 256) * (b // 256) >= 128)          # (routed only where the 256 x 256 tiles fill the chip)
    got = ops.gemm_tn_x3(dY, X)
    idx = torch.randint(0, a, (64,), generator=g).to(DEV)
    want = dY[:, idx].double().t() @ X.double()                       # 64 sampled output rows in float64
    scale = float(want.abs().max())
    assert got.shape == (a, b) and float((got[idx].double() - want).abs().max()) <= 2e-5 * scale, (shape, float((got[idx].double() - want).abs().max()), scale)
    ops.set_split3(True)
    try:
        old = ops.gemm(ops.transpose(dY), ops.transpose(X), out_dtype=torch.float32)
    finally:
        ops.set_split3(False)
    assert float((got - old).abs().max()) <= 1e-5 * scale


def test_x3_with_f16_backward_keeps_the_forward_and_gives_f16_grade_gradients(ops):
    """set_compute_dtype(float32, split3=True, backward_f16=True): the train-mode forward is the x3 forward bit for bit (logits inside the
    1e-3 clause), the backward's contractions round their operands to f16 once under the loss scale -- every head gradient within 1 % of
    its tensor's largest entry of the exact-fp32 mode's (the strict x3 backward: 2e-4), a Trainer run converges, nothing leaks out of
    the backward (the switch is off again afterwards)"""
    import sgg_amd
    from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
    from sgg_amd.trainer import Trainer
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls', min_size=S, max_size=S))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(DEV).train()
    model.dropout_p = 0.0
    batch = synthetic_batch(B=3, S=S, n_boxes=7, n_fg=3, seed=5, ragged=True)
    out, grads = {}, {}
    scale = 1024.0                                   # what the Trainer's loss scale does for the f16 operands of the backward
    for name, kw in (('exact', dict()), ('x3', dict(split3=True)), ('x3f', dict(split3=True, backward_f16=True))):
        model.load_state_dict(sd)
        model.set_compute_dtype(torch.float32, **kw)
        res = model([tuple(batch)])
        g = torch.Generator().manual_seed(0)
        Wo = torch.randn(res.rm_obj_dists.shape, generator=g).to(DEV)
        Wr = torch.randn(res.rel_dists.shape, generator=g).to(DEV)
        model.zero_grad()
        model._loss_scaled = True
        try:
            (((res.rm_obj_dists * Wo).sum() + (res.rel_dists * Wr).sum()) * scale).backward()
        finally:
            model._loss_scaled = False
        assert not ops._BWD16[0]
        out[name] = (res.rm_obj_dists.detach().clone(), res.rel_dists.detach().clone())
        grads[name] = {n: p.grad.detach().clone() / scale for n, p in model.named_parameters() if p.grad is not None}
    assert torch.equal(out['x3'][0], out['x3f'][0]) and torch.equal(out['x3'][1], out['x3f'][1])
    worst = {}
    for n, ge in grads['exact'].items():
        m = float(ge.abs().max()) + 1e-12
        worst[n] = (float((grads['x3'][n] - ge).abs().max()) / m, float((grads['x3f'][n] - ge).abs().max()) / m)
    assert max(v[0] for v in worst.values()) <= 2e-4, sorted(worst.items(), key=lambda kv: -kv[1][0])[:3]
    assert max(v[1] for v in worst.values()) <= 1e-2, sorted(worst.items(), key=lambda kv: -kv[1][1])[:3]
    assert max(v[1] for v in worst.values()) > 1e-5            # (it really is another arithmetic)
    model.load_state_dict(sd)
    model.set_compute_dtype(torch.float32, split3=True, backward_f16=True)
    tr = Trainer(model, lr=2e-2)
    losses = [float(tr.step(tuple(batch))) for _ in range(8)]
    tr.flush()
    assert losses[-1] < 0.9 * losses[0], losses
    assert all(bool(torch.isfinite(p).all()) for p in model.parameters())
    with pytest.raises(ValueError):
        model.set_compute_dtype(torch.float16, backward_f16=True)
    model.set_compute_dtype(torch.float16)
    model.eval()
    model.dropout_p = 0.5
    assert not ops.split3_on() and not model.backward_f16


def test_split3_weight_cache_follows_the_tensors_version(ops):
    """weights=True: a tensor version that comes back is split once more and then served from the cache (same object); an in-place change
    (what the optimiser's version bump announces) gets a new split; leaving the mode empties the cache"""
    w = torch.randn(96, 200, device=DEV)
    ops.split3_cache_clear()
    a = ops.split3(w, weights=True)
    b = ops.split3(w, weights=True)              # second sighting: kept from here on
    c = ops.split3(w, weights=True)
    assert c is b and torch.equal(a, b)
    w.mul_(2.0)                                  # (bumps w._version)
    d = ops.split3(w, weights=True)
    assert d is not b and torch.equal(d[:, :200].float(), w.half().float())
    act = ops.split3(w)                          # activations are never cached
    assert act is not ops.split3(w)
    ops.set_split3(True)
    ops.set_split3(False)
    assert ops.split3(w, weights=True) is not d
