"""GPU parity: every HIP kernel, called through the C ABI, against the CPU oracle on the same seeded inputs.
Integer / index work and the raster must be bit-exact; fp32 kernels within 1e-3 absolute (north_star) -- in practice
the asserted bounds are much tighter; bf16 kernels are checked against the oracle fed with bf16-rounded operands."""
import numpy as np
import pytest
import torch

from oracle import sgg_oracle as O
from tests.conftest import weights

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from sgg_amd import ops as _ops
    return _ops


def cu(x, dtype=None):
    t = torch.as_tensor(x)
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV).contiguous()


def rand_boxes(rng, n):
    xy = rng.uniform(0, 400, size=(n, 2))
    wh = rng.uniform(12, 192, size=(n, 2))
    return np.concatenate((xy, np.minimum(xy + wh, 591)), 1).astype(np.float32)


# ----------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (256, 256, 512), (200, 151, 512), (37, 51, 128), (1000, 1536, 512),
                                   (130, 4096, 1024), (5, 64, 64),
                                   # short M, long K -> split-K path
                                   (256, 4096, 8192), (37, 512, 4096), (200, 1536, 6400),
                                   # 256x256 ping-pong kernel (N >= 256 and >= 128 tiles): ragged M/N tails, 1..many K-tiles
                                   (4096, 2048, 512), (5000, 1800, 192), (4100, 2048, 128), (4096, 2050, 64), (7936, 4096, 1024)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_gemm(ops, M, N, K, dtype):
    g = torch.Generator().manual_seed(M * N + K)
    A = torch.randn(M, K, generator=g).to(dtype)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype)
    b = torch.randn(N, generator=g)
    ref = (A.float() @ W.float().t() + b).relu()
    out = ops.gemm(cu(A), cu(W), cu(b), act=ops.ACT_RELU, out_dtype=torch.float32)
    tol = 2e-5 if dtype == torch.float32 else 2e-3
    torch.testing.assert_close(out.cpu(), ref, atol=tol * max(1.0, float(ref.abs().max())), rtol=0)


def test_gemm_transpose_detecting(ops):
    # A = I pattern with an asymmetric W: catches a transposed C write
    M = N = K = 128
    A = torch.eye(M, K)
    W = torch.arange(N * K, dtype=torch.float32).reshape(N, K) / (N * K)
    out = ops.gemm(cu(A), cu(W), out_dtype=torch.float32)
    torch.testing.assert_close(out.cpu(), W.t().contiguous(), atol=1e-6, rtol=0)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_gemm_split_k_and_affine(ops, dtype):
    g = torch.Generator().manual_seed(5)
    M, K1, K2, N = 300, 256, 128, 192
    A1, A2 = torch.randn(M, K1, generator=g).to(dtype), torch.randn(M, K2, generator=g).to(dtype)
    W = (torch.randn(N, K1 + K2, generator=g) / 20).to(dtype)
    b, sc, sh = torch.randn(N, generator=g), torch.rand(N, generator=g) + .5, torch.randn(N, generator=g)
    ref = (torch.cat((A1, A2), 1).float() @ W.float().t() + b).relu() * sc + sh
    out = ops.gemm(cu(A1), cu(W), cu(b), act=ops.ACT_RELU, out_dtype=torch.float32, A2=cu(A2), post_scale=cu(sc),
                   post_shift=cu(sh))
    torch.testing.assert_close(out.cpu(), ref, atol=1e-4 if dtype == torch.float32 else 3e-2, rtol=0)
    outb = ops.gemm(cu(A1), cu(W), cu(b), act=ops.ACT_RELU, out_dtype=dtype, A2=cu(A2), post_scale=cu(sc),
                    post_shift=cu(sh))
    torch.testing.assert_close(outb.float().cpu(), ref, atol=1e-4 if dtype == torch.float32 else 6e-2, rtol=0)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_gemm_split_weights_w2(ops, dtype):
    g = torch.Generator().manual_seed(8)
    M, K1, K2, N = 5000, 1024, 512, 2048          # big enough for the 256x256 kernel too
    A1, A2 = torch.randn(M, K1, generator=g).to(dtype), torch.randn(M, K2, generator=g).to(dtype)
    W1, W2 = (torch.randn(N, K1, generator=g) / 30).to(dtype), (torch.randn(N, K2, generator=g) / 30).to(dtype)
    ref = A1.float() @ W1.float().t() + A2.float() @ W2.float().t()
    for rows in (M, 200):                         # ping-pong kernel / 128x128 kernel
        out = ops.gemm(cu(A1[:rows]), cu(W1), out_dtype=torch.float32, A2=cu(A2[:rows]), W2=cu(W2))
        torch.testing.assert_close(out.cpu(), ref[:rows], atol=2e-4 if dtype == torch.float32 else 8e-2, rtol=0)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_transpose_add_and_group_sum(ops, dtype):
    g = torch.Generator().manual_seed(2)
    R, C, group = 203, 7 * 24, 7
    x = torch.randn(R, C, generator=g).to(dtype)
    add = torch.randn(R, C // group, generator=g)
    t = ops.transpose(cu(x))
    assert tuple(t.shape) == (C, 256) and float(t[:, R:].float().abs().max()) == 0
    torch.testing.assert_close(t[:, :R].float().cpu(), x.float().t(), atol=0, rtol=0)
    t2, cs = ops.transpose(cu(x), want_colsum=True)
    torch.testing.assert_close(t2[:, :R].float().cpu(), x.float().t(), atol=0, rtol=0)
    torch.testing.assert_close(cs.cpu(), x.float().sum(0), atol=1e-3, rtol=1e-4)
    ta = ops.transpose(cu(x), add=cu(add), group=group)
    exp = (x.float() + add.repeat_interleave(group, 1)).t()
    torch.testing.assert_close(ta[:, :R].float().cpu(), exp.to(dtype).float(), atol=1e-6 if dtype == torch.float32 else 2e-2, rtol=0)
    # strided input (a column block of a wider matrix)
    wide = torch.randn(R, C + 40, generator=g).to(dtype)
    tv = ops.transpose(cu(wide)[:, 8:8 + C])
    torch.testing.assert_close(tv[:, :R].float().cpu(), wide[:, 8:8 + C].float().t(), atol=0, rtol=0)
    # odd sizes on both axes (scalar edges), more than one 128-tile each way, and the casting form f32 -> bf16
    xo = torch.randn(300, 515, generator=g).to(dtype)
    to, cso = ops.transpose(cu(xo), want_colsum=True)
    assert tuple(to.shape) == (515, 320) and float(to[:, 300:].float().abs().max()) == 0
    torch.testing.assert_close(to[:, :300].float().cpu(), xo.float().t(), atol=0, rtol=0)
    torch.testing.assert_close(cso.cpu(), xo.float().sum(0), atol=2e-3, rtol=1e-4)
    tc = ops.transpose(cu(xo.float()), dtype=torch.bfloat16)
    assert tc.dtype == torch.bfloat16
    torch.testing.assert_close(tc[:, :300].cpu(), xo.float().t().to(torch.bfloat16), atol=0, rtol=0)
    w = torch.randn(37, C, generator=g)
    gs = ops.group_sum(cu(w), C // group, group, dtype)
    torch.testing.assert_close(gs.float().cpu(), w.view(37, C // group, group).sum(2).to(dtype).float(),
                               atol=1e-5 if dtype == torch.float32 else 3e-2, rtol=0)


def test_gemm_rejects_bad_k(ops):
    with pytest.raises(ValueError):
        ops.gemm(cu(torch.zeros(8, 40)), cu(torch.zeros(8, 40)))


# ----------------------------------------------------------------------------------------- conv / pool / prep
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('B,H,W,Cin,Cout', [(1, 16, 16, 64, 64), (2, 10, 14, 64, 128), (1, 38, 38, 128, 256),
                                            (2, 76, 76, 64, 1024), (3, 62, 70, 128, 512),
                                            # LDS-resident-patch kernel (H,W >= 64): ragged tiles, 1-2 channel slabs, 64/128/256 outputs
                                            (1, 128, 128, 64, 64), (2, 130, 152, 64, 128), (1, 144, 135, 128, 256)])
def test_conv3x3(ops, dtype, B, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g).to(dtype)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(dtype)
    b = torch.randn(Cout, generator=g)
    ref = torch.nn.functional.conv2d(x.float(), w.float(), b, padding=1).relu()
    xp = torch.zeros(B, H + 2, W + 2, Cin, dtype=dtype)
    xp[:, 1:-1, 1:-1] = x.permute(0, 2, 3, 1)
    for op in (0, 1):
        out = torch.full((B, H + 2 * op, W + 2 * op, Cout), 7.0, dtype=dtype, device=DEV)
        ops.conv3x3_relu(cu(xp), cu(w.permute(0, 2, 3, 1)), cu(b), out, op)
        got = out.float().cpu()
        inner = got[:, op:H + op, op:W + op] if op else got
        tol = 1e-4 if dtype == torch.float32 else 3e-2
        torch.testing.assert_close(inner.permute(0, 3, 1, 2), ref, atol=tol, rtol=tol)
        if op:  # border untouched
            assert float(got[:, 0].min()) == 7.0 and float(got[:, :, 0].min()) == 7.0


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('nq,tw', [(42, 16), (42, 32), (22, 16), (22, 32), (24, 16), (24, 32)])
@pytest.mark.parametrize('B,H,W,Cin,Cout,pool', [(2, 70, 90, 192, 256, False), (1, 64, 64, 64, 256, True), (3, 66, 100, 64, 512, True),
                                                 (1, 97, 65, 320, 256, False)])
def test_conv3x3_pingpong_patch_kernel(ops, dtype, nq, tw, B, H, W, Cin, Cout, pool, monkeypatch):
    """csrc/conv_pp.hip under its three workgroup forms (256 px x 256 ch, 256 x 128, 512 x 128) and both tile widths (16 / 32): 2-10 channel slabs of 32 (the patch
    double buffer and the weight ring wrap several times), tiles that overhang the right and the bottom edge, with and without the fused
    pool, both output paddings; against torch's fp32 convolution of the same 16-bit operands."""
    monkeypatch.setenv('SGG_CONV_PP', str(nq))
    monkeypatch.setenv('SGG_CONV_PP_TW', str(tw))
    g = torch.Generator().manual_seed(Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g).to(dtype)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(dtype)
    b = torch.randn(Cout, generator=g)
    ref = torch.nn.functional.conv2d(x.float(), w.float(), b, padding=1).relu()
    if pool:
        ref = torch.nn.functional.max_pool2d(ref, 2)
    Ho, Wo = ref.shape[2], ref.shape[3]
    xp = torch.zeros(B, H + 2, W + 2, Cin, dtype=dtype)
    xp[:, 1:-1, 1:-1] = x.permute(0, 2, 3, 1)
    for op in (0, 1):
        out = torch.full((B, Ho + 2 * op, Wo + 2 * op, Cout), 7.0, dtype=dtype, device=DEV)
        ops.conv3x3_relu(cu(xp), cu(w.permute(0, 2, 3, 1)), cu(b), out, op, pool=pool)
        got = out.float().cpu()
        inner = got[:, op:Ho + op, op:Wo + op] if op else got
        torch.testing.assert_close(inner.permute(0, 3, 1, 2), ref, atol=3e-2, rtol=3e-2)
        if op:  # border untouched
            assert float(got[:, 0].min()) == 7.0 and float(got[:, :, 0].min()) == 7.0 and float(got[:, -1].min()) == 7.0 and float(got[:, :, -1].min()) == 7.0
    # the same launch twice: identical bits (no race between the two wave groups, the DMA ring and the patch buffers)
    o1 = torch.zeros((B, Ho + 2, Wo + 2, Cout), dtype=dtype, device=DEV)
    o2 = torch.zeros_like(o1)
    ops.conv3x3_relu(cu(xp), cu(w.permute(0, 2, 3, 1)), cu(b), o1, 1, pool=pool)
    ops.conv3x3_relu(cu(xp), cu(w.permute(0, 2, 3, 1)), cu(b), o2, 1, pool=pool)
    assert torch.equal(o1, o2)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_conv3x3_relu_with_fused_maxpool(ops, dtype):
    """conv + ReLU + MaxPool2d(2) in one kernel == the two separate kernels, bit for bit (max commutes with bias + ReLU
    and with the output rounding); sizes that leave partial 16x16 tiles on both axes."""
    g = torch.Generator().manual_seed(12)
    B, H, W, Cin, Cout = 2, 74, 100, 64, 128
    x = torch.zeros(B, H + 2, W + 2, Cin)
    x[:, 1:-1, 1:-1] = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(Cout, 3, 3, Cin, generator=g) / 24
    bias = torch.randn(Cout, generator=g) * 0.2
    xd, wd, bd = cu(x.to(dtype)), cu(w.to(dtype)), cu(bias)
    full = torch.zeros(B, H + 2, W + 2, Cout, device=DEV, dtype=dtype)
    ops.conv3x3_relu(xd, wd, bd, full, 1)
    ref = torch.zeros(B, H // 2 + 2, W // 2 + 2, Cout, device=DEV, dtype=dtype)
    ops.maxpool2x2(full, ref, 1)
    got = torch.full((B, H // 2 + 2, W // 2 + 2, Cout), 7.0, device=DEV, dtype=dtype)
    ops.conv3x3_relu(xd, wd, bd, got, 1, pool=True)
    assert torch.equal(got[:, 1:-1, 1:-1], ref[:, 1:-1, 1:-1])
    assert float(got[:, 0].float().min()) == 7.0 and float(got[:, :, -1].float().min()) == 7.0     # border untouched
    assert ops.conv_pool_fusable(H, W, Cout) and not ops.conv_pool_fusable(37, 37, 512)
    with pytest.raises(ValueError):
        ops.conv3x3_relu(xd[:, :-1], wd, bd, got, 1, pool=True)                                   # odd height


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('B,H,W,pool', [(2, 64, 96, True), (1, 70, 58, True), (3, 48, 80, False), (1, 608, 608, True)])
def test_conv1_block_equals_conv1_1_then_conv1_2(ops, dtype, B, H, W, pool):
    """sgg_conv1_block (conv1_1 computed inside conv1_2's LDS patch) == the two launches, bit for bit: the same MFMA products, the same
    16-bit rounding of conv1_1's output, zero padding at the image border, tiles that overhang the right / bottom edge."""
    g = torch.Generator().manual_seed(H + W)
    img = torch.zeros(B, H + 2, W + 2, 4)
    img[:, 1:-1, 1:-1, :3] = torch.randn(B, H, W, 3, generator=g)
    w1 = torch.randn(64, 27, generator=g) / 5
    b1 = torch.randn(64, generator=g) * 0.1
    w2 = (torch.randn(64, 3, 3, 64, generator=g) / 24).to(dtype)
    b2 = torch.randn(64, generator=g) * 0.1
    imd, w1d, b1d, w2d, b2d = cu(img), cu(w1), cu(b1), cu(w2), cu(b2)
    y1 = torch.zeros(B, H + 2, W + 2, 64, device=DEV, dtype=dtype)
    ops.conv1_1(imd, w1d, b1d, y1)
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    ref = torch.full((B, Ho + 2, Wo + 2, 64), 7.0, device=DEV, dtype=dtype)
    ops.conv3x3_relu(y1, w2d, b2d, ref, 1, pool=pool)
    got = torch.full((B, Ho + 2, Wo + 2, 64), 7.0, device=DEV, dtype=dtype)
    frags = ops.conv1_pack_weights(w1d, dtype)
    ops.conv1_block(imd, frags, b1d, w2d, b2d, got, 1, pool=pool)
    assert torch.equal(got, ref)
    assert float(got[:, 1:-1, 1:-1].float().abs().max()) > 0.1 and float(got[:, 0].float().min()) == 7.0
    got0 = torch.empty((B, Ho, Wo, 64), device=DEV, dtype=dtype)
    ops.conv1_block(imd, frags, b1d, w2d, b2d, got0, 0, pool=pool)
    assert torch.equal(got0, ref[:, 1:-1, 1:-1])


def test_image_prep_conv1_1_maxpool(ops):
    g = torch.Generator().manual_seed(3)
    B, S, Hp = 2, 40, 64
    imgs = [torch.rand(3, S, S, generator=g) for _ in range(B)]
    batch, sizes, _ = O.transform(imgs, None, min_size=S, max_size=S)
    assert batch.shape[-1] == Hp
    buf = torch.zeros(B, Hp + 2, Hp + 2, 4, device=DEV)
    for b, im in enumerate(imgs):
        ops.image_prep(cu(im), S, S, buf, b)
    torch.testing.assert_close(buf[:, 1:-1, 1:-1, :3].permute(0, 3, 1, 2).cpu(), batch, atol=1e-6, rtol=1e-6)
    assert float(buf[..., 3].abs().max()) == 0
    # resize path (20 -> 40)
    small = torch.rand(3, 20, 20, generator=g)
    b2, _, _ = O.transform([small], None, min_size=S, max_size=S)
    buf2 = torch.zeros(1, Hp + 2, Hp + 2, 4, device=DEV)
    ops.image_prep(cu(small), S, S, buf2, 0)
    torch.testing.assert_close(buf2[:, 1:-1, 1:-1, :3].permute(0, 3, 1, 2).cpu(), b2, atol=1e-5, rtol=1e-5)
    # conv1_1
    w = torch.randn(64, 3, 3, 3, generator=g) * 0.2
    bias = torch.randn(64, generator=g) * 0.1
    ref = torch.nn.functional.conv2d(batch, w, bias, padding=1).relu()
    out = torch.zeros(B, Hp + 2, Hp + 2, 64, device=DEV)
    ops.conv1_1(buf, cu(w.permute(0, 2, 3, 1).reshape(64, 27)), cu(bias), out)
    torch.testing.assert_close(out[:, 1:-1, 1:-1].permute(0, 3, 1, 2).cpu(), ref, atol=1e-5, rtol=1e-5)
    # conv1_1, bf16 storage mode = MFMA kernel (operands rounded to bf16, fp32 accumulate)
    outb = torch.zeros(B, Hp + 2, Hp + 2, 64, device=DEV, dtype=torch.bfloat16)
    ops.conv1_1(buf, cu(w.permute(0, 2, 3, 1).reshape(64, 27)), cu(bias), outb)
    refb = torch.nn.functional.conv2d(batch.bfloat16().float(), w.bfloat16().float(), bias, padding=1).relu()
    torch.testing.assert_close(outb[:, 1:-1, 1:-1].float().permute(0, 3, 1, 2).cpu(), refb, atol=2e-2, rtol=2e-2)
    assert float(outb[:, 0].float().abs().max()) == 0 and float(outb[:, :, -1].float().abs().max()) == 0
    # maxpool
    pooled = torch.zeros(B, Hp // 2 + 2, Hp // 2 + 2, 64, device=DEV)
    ops.maxpool2x2(out, pooled, 1)
    torch.testing.assert_close(pooled[:, 1:-1, 1:-1].permute(0, 3, 1, 2).cpu(), torch.nn.functional.max_pool2d(ref, 2),
                               atol=1e-5, rtol=1e-5)
    assert float(pooled[:, 0].abs().max()) == 0


# ----------------------------------------------------------------------------------------- pairs (bit-exact)
@pytest.mark.parametrize('tag', ['b1', 'b3', 'b8x32'])
def test_pair_index_matches_reference_golden(ops, golden, tag):
    g = golden('pairs')
    im, boxes = cu(g[tag + '_im_inds']), cu(g[tag + '_boxes'])
    for ov in (0, 1):
        out, cnt = ops.pair_index_eval(im, boxes, bool(ov))
        n = int(cnt.item())
        np.testing.assert_array_equal(out[:n].cpu().numpy(), g['%s_eval_ov%d' % (tag, ov)])
    imn = g[tag + '_im_inds']
    first = np.array([np.argmax(imn == i) for i in range(imn.max() + 1)], np.int32)
    exp = g[tag + '_train_rel_labels']
    out, cnt = ops.pair_index_train(im, cu(g[tag + '_gt_rels']), cu(first), len(exp) + 5)
    assert int(cnt.item()) == len(exp)
    np.testing.assert_array_equal(out[:len(exp)].cpu().numpy(), exp)


def test_pair_index_ragged_large_and_capacity(ops):
    rng = np.random.RandomState(0)
    sizes = [1, 70, 3, 129, 2]
    im = np.concatenate([np.full(n, i, np.int64) for i, n in enumerate(sizes)])
    boxes = rand_boxes(rng, len(im))
    for ov in (False, True):
        exp = O.get_rel_inds_eval(im, boxes, ov)
        out, cnt = ops.pair_index_eval(cu(im), cu(boxes), ov)
        assert int(cnt.item()) == len(exp)
        np.testing.assert_array_equal(out[:len(exp)].cpu().numpy(), exp)
    # capacity smaller than needed: count still reported, no overflow write
    out, cnt = ops.pair_index_eval(cu(im), None, False, cap=10)
    assert int(cnt.item()) == sum(n * (n - 1) for n in sizes) and out.shape[0] == 10
    # duplicate FG relation on one pair -> two rows
    gt_rels = np.array([[1, 0, 1, 5], [1, 0, 1, 9], [3, 2, 0, 7]], np.int64)
    first = np.cumsum([0] + sizes[:-1]).astype(np.int32)
    rois = np.concatenate((im[:, None].astype(np.float32), boxes), 1)
    _, _, exp = O.proposal_assignments_gtbox(rois, boxes, np.stack((im, im), 1), gt_rels, RELS_PER_IMG=100000)
    out, cnt = ops.pair_index_train(cu(im), cu(gt_rels), cu(first), len(exp))
    assert int(cnt.item()) == len(exp)
    np.testing.assert_array_equal(out.cpu().numpy(), exp)


def test_edge_csr(ops):
    rng = np.random.RandomState(1)
    im = np.concatenate([np.full(n, i, np.int64) for i, n in enumerate([5, 9, 1, 4])])
    rel = O.get_rel_inds_eval(im)
    rel = rel[rng.rand(len(rel)) > 0.3]
    N = len(im)
    for im_arg in (None, cu(im)):      # full scan / image-segment scan (rel sorted by image)
        optr, oids, iptr, iids, so, flags = [t.cpu().numpy() for t in ops.edge_csr(cu(rel), N, im_arg)]
        np.testing.assert_array_equal(so[:len(rel)], rel[:, 1:])
        assert flags[0] == 1                        # rel is sorted by subject
        for n in range(N):
            np.testing.assert_array_equal(oids[optr[n]:optr[n + 1]], np.nonzero(rel[:, 1] == n)[0])
            np.testing.assert_array_equal(iids[iptr[n]:iptr[n + 1]], np.nonzero(rel[:, 2] == n)[0])
        assert optr[N] == iptr[N] == len(rel)


# ----------------------------------------------------------------------------------------- raster (bit-exact)
def test_union_rects_bit_exact_vs_reference_golden(ops, golden):
    g = golden('raster')
    pairs8 = g['pairs']
    E = len(pairs8)
    rois = np.concatenate((np.zeros((2 * E, 1), np.float32), pairs8.reshape(2 * E, 4)), 1)
    pidx = np.arange(2 * E, dtype=np.int64).reshape(E, 2)
    out = ops.union_rects(cu(rois), cu(pidx), 27, 0.0)
    np.testing.assert_array_equal(out.cpu().numpy(), g['out27'])
    out7 = ops.union_rects(cu(rois), cu(pidx[:16]), 7, 0.0)
    np.testing.assert_array_equal(out7.cpu().numpy(), g['out7'])
    # with the caller's -0.5 (lib/get_union_boxes.py:67)
    outm = ops.union_rects(cu(rois), cu(pidx), 27, -0.5)
    np.testing.assert_array_equal(outm.cpu().numpy(), g['out27'] - np.float32(0.5))


def test_union_rects_full_size_and_patches(ops):
    rng = np.random.RandomState(2)
    N = 32
    boxes = rand_boxes(rng, N)
    rois = np.concatenate((np.zeros((N, 1), np.float32), boxes), 1)
    pairs = O.get_rel_inds_eval(np.zeros(N, np.int64))[:, 1:]
    pr = np.concatenate((boxes[pairs[:, 0]], boxes[pairs[:, 1]]), 1)
    exp = O.draw_union_boxes(pr, 27) - np.float32(0.5)
    out = ops.union_rects(cu(rois), cu(pairs), 27, -0.5).cpu().numpy()
    np.testing.assert_array_equal(out, exp)
    # patches: rows {-3..3} and {13..19} with zero padding
    pat = ops.union_rect_patches(cu(rois), cu(pairs), torch.float32).cpu().numpy().reshape(len(pairs), 4, 128)
    padded = np.zeros((len(pairs), 2, 27 + 6, 27 + 6), np.float32)
    padded[:, :, 3:30, 3:30] = exp
    for pos in range(4):
        oy, ox = pos >> 1, pos & 1
        win = padded[:, :, oy * 16:oy * 16 + 7, ox * 16:ox * 16 + 7].reshape(len(pairs), 98)
        np.testing.assert_array_equal(pat[:, pos, :98], win)
    assert np.abs(pat[:, :, 98:]).max() == 0


def test_degenerate_union_matches_reference_nan_behaviour(ops):
    rois = np.array([[0, 5, 5, 5, 9], [0, 5, 6, 5, 8]], np.float32)
    out = ops.union_rects(cu(rois), cu(np.array([[0, 1]], np.int64)), 27, 0.0).cpu().numpy()
    exp = O.draw_union_boxes(np.array([[5, 5, 5, 9, 5, 6, 5, 8]], np.float32), 27)
    np.testing.assert_array_equal(np.isnan(out), np.isnan(exp))
    np.testing.assert_array_equal(np.nan_to_num(out, nan=-7), np.nan_to_num(exp, nan=-7))


@pytest.mark.parametrize('tag', ['d128', 'd32'])
def test_rect_feat_pipeline_vs_reference_golden(ops, golden, tag):
    """raster patches -> GEMM(+ReLU+BN) -> max4 -> GEMM(+ReLU+BN) == UnionBoxesAndFeats.conv of the reference."""
    g = golden('union_feats')
    from sgg_amd.union_boxes import fold_rect_conv, rect_feat
    p = {k: cu(v) for k, v in weights(g, tag).items()}
    prep = fold_rect_conv(p, torch.float32)
    rf = rect_feat(cu(g[tag + '_rois']), cu(g[tag + '_union_inds']), prep, torch.float32)
    np.testing.assert_allclose(rf.cpu().numpy(), g[tag + '_rect_feat'], atol=2e-5)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_roi_align_bwd_is_the_adjoint_of_the_forward(ops, dtype):
    """sgg_roi_align_bwd: <g, RoIAlign(f)> == <RoIAlign^T(g), f> for random f, g (node boxes and fused union boxes, incl. boxes that
    leave the map and a degenerate one) -- the forward itself is checked against the oracle in test_roi_align; plus a finite-
    difference probe of single feature-map entries."""
    rng = np.random.RandomState(9)
    B, C, H, W, N = 2, 64, 20, 24, 9
    f = torch.from_numpy(rng.randn(B, H, W, C).astype(np.float32))
    xy = rng.uniform(0, 250, size=(N, 2))
    boxes = np.concatenate((xy, xy + rng.uniform(12, 180, size=(N, 2))), 1).astype(np.float32)    # some stick out of the 384 x 320 image
    boxes[0] = [50, 50, 50, 50]
    im = np.sort(rng.randint(0, B, N)).astype(np.float32)
    rois = np.concatenate((im[:, None], boxes), 1)
    pairs = np.array([(i, j) for i in range(N) for j in range(N) if i != j and im[i] == im[j]], np.int64)
    fd = cu(f)
    for pr in (None, cu(pairs)):
        y = ops.roi_align(fd, cu(rois), pr).float()
        g = torch.from_numpy(rng.randn(*y.shape).astype(np.float32)).to(dtype)
        gf = ops.roi_align_bwd(cu(g), (B, H, W, C), cu(rois), pr)
        lhs = float((y.double().cpu() * g.double()).sum())
        rhs = float((gf.double().cpu() * f.double()).sum())
        assert abs(lhs - rhs) <= 2e-4 * (abs(lhs) + float((y.abs().double().cpu() * g.abs().double()).sum()) * 1e-3 + 1), (lhs, rhs)
        # d <g, y> / d f[b,y,x,c] by central differences (RoIAlign is linear in f: the difference quotient is exact up to rounding)
        for _ in range(3):
            b_, y_, x_, c_ = int(rng.randint(B)), int(rng.randint(H)), int(rng.randint(W)), int(rng.randint(C))
            fp = f.clone(); fp[b_, y_, x_, c_] += 1.0
            yp = ops.roi_align(cu(fp), cu(rois), pr).float()
            fdq = float(((yp - y).double().cpu() * g.double()).sum())
            assert abs(fdq - float(gf[b_, y_, x_, c_])) <= 1e-3 * (1 + abs(fdq)), (fdq, float(gf[b_, y_, x_, c_]))
    acc = ops.roi_align_bwd(cu(g), (B, H, W, C), cu(rois), cu(pairs), d_fmap=gf.clone())      # accumulates into a given map
    torch.testing.assert_close(acc, 2 * gf, rtol=1e-5, atol=1e-5)


def test_roi_align_bwd_equals_the_transposed_oracle_operator(ops):
    """An oracle gradient for sgg_roi_align_bwd that does not come from the HIP forward: RoIAlign is linear in the feature map, so
    the ORACLE's forward applied to the unit maps gives the operator A column by column (one 6x7 map, [R*49, 42]); the gradient is
    A^T g.  Node boxes and fused union boxes; boxes that leave the map, a degenerate box, a box over the whole map."""
    rng = np.random.RandomState(31)
    H, W, C = 6, 7, 8
    boxes = np.array([[10, 8, 70, 60], [0, 0, 111, 95], [40, 40, 40, 40], [60, 30, 150, 130], [-20, -10, 30, 50]], np.float32)
    rois = np.concatenate((np.zeros((len(boxes), 1), np.float32), boxes), 1)
    pairs = np.array([(0, 1), (1, 0), (2, 3), (4, 0), (3, 4)], np.int64)
    union = np.concatenate((np.zeros((len(pairs), 1), np.float32), np.minimum(boxes[pairs[:, 0], :2], boxes[pairs[:, 1], :2]),
                            np.maximum(boxes[pairs[:, 0], 2:], boxes[pairs[:, 1], 2:])), 1)
    for pr, rr in ((None, rois), (cu(pairs), union)):
        R = len(rr)
        A = np.zeros((R * 49, H * W), np.float64)
        for k in range(H * W):
            e = np.zeros((1, 1, H, W), np.float32)
            e.reshape(-1)[k] = 1.0
            A[:, k] = O.roi_align(e, rr).reshape(-1)
        g = rng.randn(R, C, 7, 7).astype(np.float32)
        want = np.einsum('rk,rc->kc', A, g.transpose(0, 2, 3, 1).reshape(R * 49, C).astype(np.float64)).reshape(1, H, W, C)
        got = ops.roi_align_bwd(cu(torch.from_numpy(g)), (1, H, W, C), cu(rois), pr).double().cpu().numpy()
        np.testing.assert_allclose(got, want, atol=2e-5 * max(1.0, np.abs(want).max()))


def test_raw_boxes_raster_and_rect_feat_vs_reference_golden(ops, golden):
    """edge_model 'raw_boxes' (lib/get_union_boxes.py:69-116): the grid_sample raster, the patches the conv stack reads from it,
    and the module's eval forward, against vectors from the reference module."""
    g = golden('union_feats')
    ims = [tuple(int(v) for v in x) for x in g['raw_im_sizes']]
    rois, ui = cu(g['raw_rois']), cu(g['raw_union_inds'])
    r = ops.union_rects(rois, ui, 27, 0.0, im_sizes=ims)
    np.testing.assert_allclose(r.cpu().numpy(), g['raw_rects'], atol=1e-5)
    np.testing.assert_allclose(ops.union_rects(rois, ui, 27, -0.5, im_sizes=ims).cpu().numpy(), g['raw_rects'] - 0.5, atol=1e-5)
    # patches = the 4 stride-16 7x7 windows of (raster - 0.5), zero outside
    pat = ops.union_rect_patches(rois, ui, torch.float32, 27, 128, im_sizes=ims).cpu().numpy().reshape(len(g['raw_union_inds']), 4, 128)
    padded = np.zeros((len(pat), 2, 27 + 6 + 16, 27 + 6 + 16), np.float32)
    padded[:, :, 3:30, 3:30] = g['raw_rects'] - 0.5
    for pos in range(4):
        oy, ox = (pos >> 1) * 16, (pos & 1) * 16
        np.testing.assert_allclose(pat[:, pos, :98], padded[:, :, oy:oy + 7, ox:ox + 7].reshape(len(pat), 98), atol=1e-5)
    assert np.abs(pat[:, :, 98:]).max() == 0
    from sgg_amd.union_boxes import UnionBoxesAndFeats
    m = UnionBoxesAndFeats(pooling_size=7, stride=16, dim=32, edge_model='raw_boxes')
    m.load_state_dict({k[len('raw_w_'):]: torch.from_numpy(g[k]) for k in g if k.startswith('raw_w_')}, strict=False)
    m.to(DEV).eval()
    out = m(cu(g['raw_pools']), rois, ui, ims)
    np.testing.assert_allclose(out.cpu().numpy(), g['raw_out'], atol=5e-5)
    with pytest.raises(ValueError):
        m(cu(g['raw_pools']), rois, ui, None)                 # the image sizes are part of this raster
    with pytest.raises(NotImplementedError):
        UnionBoxesAndFeats(edge_model='boxes')


# ----------------------------------------------------------------------------------------- RoIAlign
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_roi_align(ops, dtype):
    rng = np.random.RandomState(4)
    B, C, H, W, N = 3, 64, 38, 38, 12
    fm = torch.from_numpy(rng.randn(B, C, H, W).astype(np.float32)).to(dtype)
    boxes = rand_boxes(rng, N)
    boxes[0] = [80, 80, 80, 80]            # degenerate -> min size 1
    boxes[1] = [560, 560, 607, 607]        # touches the map edge
    im = np.sort(rng.randint(0, B, N)).astype(np.float32)
    rois = np.concatenate((im[:, None], boxes), 1)
    pairs = np.array([(i, j) for i in range(N) for j in range(N) if i != j and im[i] == im[j]], np.int64)
    exp_n, exp_e = O.node_edge_features(fm.float().numpy(), rois, pairs)
    fm_nhwc = cu(fm.permute(0, 2, 3, 1))
    out_n = ops.roi_align(fm_nhwc, cu(rois))
    assert tuple(out_n.shape) == (N, C, 7, 7) and out_n.is_contiguous()      # the reference's layout
    got_n = out_n.float().cpu().numpy()
    got_e = ops.roi_align(fm_nhwc, cu(rois), cu(pairs)).float().cpu().numpy()
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    np.testing.assert_allclose(got_n, exp_n, atol=tol)
    np.testing.assert_allclose(got_e, exp_e, atol=tol)
    if dtype == torch.float32:  # fused broadcast add
        add = torch.from_numpy(rng.randn(len(pairs), C).astype(np.float32))
        got = ops.roi_align(fm_nhwc, cu(rois), cu(pairs), add_ec=cu(add)).cpu().numpy()
        np.testing.assert_allclose(got, exp_e + add.numpy()[:, :, None, None], atol=tol)
        x = cu(torch.from_numpy(exp_e)).reshape(len(pairs), C, 49).contiguous()
        ops.bcast_add_(x, cu(add))
        np.testing.assert_allclose(x.cpu().numpy().reshape(len(pairs), C, 7, 7), exp_e + add.numpy()[:, :, None, None],
                                   atol=1e-6)


# ----------------------------------------------------------------------------------------- IMP + GRU
def test_gru_gate_vs_reference_golden(ops, golden):
    g = golden('gru')
    w = {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('w_')}
    x, h = torch.from_numpy(g['x']), torch.from_numpy(g['h'])
    gi = x @ w['weight_ih'].t() + w['bias_ih']
    gh = h @ w['weight_hh'].t() + w['bias_hh']
    out = ops.gru_gate(cu(gi), cu(gh), None, cu(h), torch.float32)
    np.testing.assert_allclose(out.cpu().numpy(), g['out'], atol=2e-6)
    out0 = ops.gru_gate(cu(gi), None, cu(w['bias_hh']), None, torch.float32)
    np.testing.assert_allclose(out0.cpu().numpy(), g['out_h0'], atol=2e-6)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize('H', [8, 16, 32, 64, 128, 256, 512])
def test_gru_gate_dot_epilogue_every_row_width(ops, dtype, H):
    """the four gate dot products a gate kernel emits = (the state row AS STORED) . w_k, for every row width the DPP reduction handles
    (H / 8 lanes per row: 1 .. 64; several rows per wave below 512)"""
    g = torch.Generator().manual_seed(H)
    M = 77
    gi, gh, hp = torch.randn(M, 3 * H, generator=g), torch.randn(M, 3 * H, generator=g), torch.randn(M, H, generator=g)
    gate_w = cu(torch.randn(4, 2 * H, generator=g))
    for dot_w in (gate_w[:, :H], gate_w[:, H:]):
        out, dots = ops.gru_gate(cu(gi), cu(gh), None, cu(hp.to(dtype)), dtype, dot_w=dot_w)
        want = out.double() @ dot_w.double().t()
        assert float((dots.double() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))


def test_gate_dots_are_reproducible_beside_another_streams_gemm(ops):
    """Round 5: the x3 mode's logits moved from run to run because the f32 gate kernel's dot products -- then a
    ds_bpermute butterfly -- came out wrong in ~15 % of its launches while the node lane ran the x3 GEMM of the node GRU (tools/gate_race.py).
    The same arrangement here: 200 launches beside that GEMM, every one bit-equal to the first and right."""
    torch.manual_seed(0)
    E, N, H = 7936, 256, 512
    gi, b_hh = torch.randn(E, 3 * H, device=DEV), torch.randn(3 * H, device=DEV) * 0.1
    gate_w = (torch.randn(4, 2 * H, device=DEV) * 0.05).contiguous()
    we = gate_w[:, H:]
    obj, W = torch.randn(N, H, device=DEV), (torch.randn(3 * H, H, device=DEV) * 0.04).contiguous()
    side = torch.cuda.Stream(device=DEV)
    ref_h, ref_d = ops.gru_gate(gi, None, b_hh, None, torch.float32, dot_w=we)
    assert float((ref_d - ref_h @ we.t()).abs().max()) < 1e-5
    bad = 0
    prev = ops.set_split3(False)
    try:
        for _ in range(200):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                ops.set_split3(True)
                for _ in range(3):
                    ops.gemm(obj, W, None, out_dtype=torch.float32)
                ops.set_split3(False)
            h, d = ops.gru_gate(gi, None, b_hh, None, torch.float32, dot_w=we)
            bad += int(not (torch.equal(d, ref_d) and torch.equal(h, ref_h)))
            torch.cuda.current_stream().wait_stream(side)
    finally:
        ops.set_split3(prev)
    torch.cuda.synchronize()
    assert bad == 0, '%d of 200 launches differ' % bad


@pytest.mark.parametrize('tag', ['h32_b1', 'h64_b3', 'h32_sampled', 'h128_b2'])
def test_message_pass_vs_reference_golden(ops, golden, tag):
    g = golden('message_pass')
    from sgg_amd.imp import ImpWeights, message_pass
    p = {k: cu(v) for k, v in weights(g, tag).items()}
    wts = ImpWeights.from_state(p, torch.float32)
    rel = cu(g[tag + '_rel_inds'])
    N = g[tag + '_obj_rep'].shape[0]
    csr = ops.edge_csr(rel, N)
    for it in range(4):
        v, e = message_pass(cu(g[tag + '_rel_rep']), cu(g[tag + '_obj_rep']), rel, csr, wts, it, torch.float32)
        np.testing.assert_allclose(v.cpu().numpy(), g['%s_v%d' % (tag, it)], atol=3e-5)
        np.testing.assert_allclose(e.cpu().numpy(), g['%s_e%d' % (tag, it)], atol=3e-5)


def _gates(v, e, rel, gw, gb):
    """the four gates of every edge, dense restatement of rel_model_stanford.py:78-89 (f32 torch tensors on any device)"""
    s, o = rel[:, 1], rel[:, 2]
    sv, ov = v[s], v[o]
    return [torch.sigmoid(torch.cat((a, e), 1) @ gw[k] + gb[k]) for k, a in enumerate((sv, ov, sv, ov))]


def _ctx_expect(x, rel, ga, gbv, N):
    """out_a[n] = sum_{s(e)=n} ga(e) x[e], out_b[n] = sum_{o(e)=n} gb(e) x[e]  (the one-hot matmuls of rel_model_stanford.py:60-66, 91)"""
    s, o = rel[:, 1], rel[:, 2]
    z = torch.zeros((N, x.shape[1]), dtype=torch.float32, device=x.device)
    return z.clone().index_add_(0, s, ga[:, None] * x), z.clone().index_add_(0, o, gbv[:, None] * x)


def _states(ops, g, M, H, dtype, dot_w):
    """a state tensor and its gate dot products, both made by the GRU gate kernel (its dot epilogue is what feeds the IMP kernels)"""
    gi, gh = torch.randn(M, 3 * H, generator=g), torch.randn(M, 3 * H, generator=g)
    hp = torch.randn(M, H, generator=g)
    out, dots = ops.gru_gate(cu(gi), cu(gh), None, cu(hp.to(dtype)), dtype, dot_w=dot_w)
    plain = ops.gru_gate(cu(gi), cu(gh), None, cu(hp.to(dtype)), dtype)
    assert torch.equal(out, plain)                                   # the epilogue does not change the state
    return out, dots


TOL16 = {torch.float32: (dict(atol=2e-5, rtol=1e-5), dict(atol=1e-4, rtol=1e-5)),
         torch.bfloat16: (dict(atol=3e-2, rtol=1e-2), dict(atol=0.3, rtol=2e-2)),
         torch.float16: (dict(atol=4e-3, rtol=2e-3), dict(atol=4e-2, rtol=3e-3))}


@pytest.mark.parametrize('H,sizes,dtype', [
    (512, [32, 32], torch.float32),              # benchmark graphs, f32: 16-channel slices
    (512, [32, 7, 1, 20, 2], torch.float32),     # ragged batch incl. a graph without edges; sampled edge list
    (512, [32] * 3, torch.bfloat16),             # bf16: 32-channel slices (LP = 4)
    (512, [32] * 3, torch.float16),
    (512, [40, 36], torch.bfloat16),             # 1560 edges per graph
    (512, [52, 5], torch.float16),               # 2652 edges: 32-byte pieces (LP = 2)
    (64, [9, 32, 4], torch.float32),             # narrow rows (golden-sized hidden dim)
    (32, [6, 6], torch.bfloat16),                # 64-byte rows: one LP = 4 slice
    (1536, [32, 11], torch.bfloat16),            # the backward's use: rows of d_gi (3H channels)
])
def test_imp_ctx_vs_dense_formula(ops, H, sizes, dtype, monkeypatch):
    """sgg_imp_ctx_fwd -- every form that applies: CSR lists (any edge list), every-row-once sliced kernel, the matrix-core product --
    fed by the gate-dot epilogue of sgg_gru_gate_fwd, against the dense restatement of rel_model_stanford.py:86-91; both gate pairs
    (pair 2 = the forward's context sums, pair 0 = the gradient of the node projection), halves and summed output."""
    g = torch.Generator().manual_seed(11 + H + len(sizes))
    im = np.concatenate([np.full(n, b) for b, n in enumerate(sizes)]).astype(np.int64)
    rel = O.get_rel_inds_eval(im)
    if sizes[0] == 32 and len(sizes) == 5:       # sampled edge list: drop ~40 % of the rows, node 3 loses its out-edges
        rng = np.random.RandomState(2)
        keep = rng.rand(len(rel)) > 0.4
        keep[rel[:, 1] == 3] = False
        rel = rel[keep]
    N, E = len(im), len(rel)
    Hs = 512 if H == 1536 else H                 # the states that make the dots (the rows x may be wider: d_gi)
    gw, gb = torch.randn(4, 2 * Hs, generator=g) / (Hs ** 0.5), torch.randn(4, generator=g)
    gwd = cu(gw)
    v_d, nd = _states(ops, g, N, Hs, dtype, gwd[:, :Hs])
    e_d, ed = _states(ops, g, E, Hs, dtype, gwd[:, Hs:])
    v, e = v_d.float().cpu(), e_d.float().cpu()
    torch.testing.assert_close(nd.cpu(), v @ gw[:, :Hs].t(), atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(ed.cpu(), e @ gw[:, Hs:].t(), atol=2e-5, rtol=1e-5)
    x_d = e_d if H == Hs else cu(torch.randn(E, H, generator=g).to(dtype))
    x = x_d.float().cpu()
    per_graph = [int((rel[:, 0] == b).sum()) for b in range(len(sizes))]
    graphs = (len(sizes), max(sizes), max(per_graph))
    hint = ops.edge_csr(cu(rel), N, cu(im), graphs=graphs)
    plain = ops.edge_csr(cu(rel), N, cu(im))
    assert ops.imp_sliced_ok(hint, H, dtype) and not ops.imp_sliced_ok(plain, H, dtype)
    np.testing.assert_array_equal(hint.img_ptr.cpu().numpy()[:len(sizes) + 1], np.concatenate(([0], np.cumsum(sizes))))
    np.testing.assert_array_equal(hint.img_ptr.cpu().numpy()[len(sizes) + 1:2 * len(sizes) + 2], np.concatenate(([0], np.cumsum(per_graph))))
    gt = _gates(v, e, torch.from_numpy(rel), gw, gb)
    tol = TOL16[dtype][1]
    forms = [('lists', plain, None), ('sliced', hint, 's')]
    if dtype != torch.float32 and max(sizes) <= 32 and max(per_graph) <= 1024 and (H * 2) % 128 == 0:
        forms.append(('mfma', hint, 'm'))
    for pair in (2, 0):
        exp_a, exp_b = _ctx_expect(x, torch.from_numpy(rel), gt[pair], gt[pair + 1], N)
        for name, csr, env in forms:
            if env:
                monkeypatch.setenv('SGG_IMP_CTX', env)
            else:
                monkeypatch.delenv('SGG_IMP_CTX', raising=False)
            ctx2 = ops.imp_ctx(x_d, csr, N, nd, ed, cu(gb), pair=pair)
            csum = ops.imp_ctx(x_d, csr, N, nd, ed, cu(gb), pair=pair, ctx_sum=torch.empty((N, H), dtype=dtype, device=DEV))
            torch.testing.assert_close(ctx2[0].float().cpu(), exp_a, msg=lambda m: '%s %s pair %d: %s' % (name, 'a', pair, m), **tol)
            torch.testing.assert_close(ctx2[1].float().cpu(), exp_b, msg=lambda m: '%s %s pair %d: %s' % (name, 'b', pair, m), **tol)
            torch.testing.assert_close(csum.float().cpu(), exp_a + exp_b, msg=lambda m: '%s sum pair %d: %s' % (name, pair, m), **tol)
            again = ops.imp_ctx(x_d, csr, N, nd, ed, cu(gb), pair=pair)
            assert torch.equal(again, ctx2), name                     # no atomics: bit-reproducible
    monkeypatch.delenv('SGG_IMP_CTX', raising=False)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('H,sizes', [(512, [32, 9]), (64, [5, 1, 12])])
def test_gru_gate_proj_equals_gru_on_materialised_edge_inputs(ops, dtype, H, sizes):
    """sgg_gru_gate_proj_fwd: the edge GRU from the node projection P = v W_ih^T equals the GRU gate kernel fed with the pre-activations of
    the MATERIALISED edge inputs, gi[e] = W_ih (g_sub v[s] + g_obj v[o]) + b_ih = g_sub P[s] + g_obj P[o] + b_ih (rel_model_stanford.py:76-83);
    same dot-product epilogue."""
    g = torch.Generator().manual_seed(3 + H)
    im = np.concatenate([np.full(n, b) for b, n in enumerate(sizes)]).astype(np.int64)
    rel = O.get_rel_inds_eval(im)
    N, E = len(im), len(rel)
    P = torch.randn(N, 3 * H, generator=g)
    gh = torch.randn(E, 3 * H, generator=g)
    b_ih = torch.randn(3 * H, generator=g) / 4
    nd, ed, gb = torch.randn(N, 4, generator=g), torch.randn(E, 4, generator=g), torch.randn(4, generator=g)
    hp = torch.randn(E, H, generator=g).to(dtype)
    gw = torch.randn(4, 2 * H, generator=g) / (H ** 0.5)
    s, o = torch.from_numpy(rel[:, 1]), torch.from_numpy(rel[:, 2])
    g_sub = torch.sigmoid(nd[s, 0] + ed[:, 0] + gb[0])
    g_obj = torch.sigmoid(nd[o, 1] + ed[:, 1] + gb[1])
    gi = g_sub[:, None] * P[s] + g_obj[:, None] * P[o] + b_ih
    csr = ops.edge_csr(cu(rel), N, cu(im))
    dot_w = cu(gw)[:, H:]
    out, dots = ops.gru_gate_proj(cu(gh), cu(P), cu(b_ih), csr, cu(nd), cu(ed), cu(gb), cu(hp), dot_w=dot_w)
    ref, rdots = ops.gru_gate(cu(gi), cu(gh), None, cu(hp), dtype, dot_w=dot_w)
    tol = dict(atol=3e-6, rtol=1e-5) if dtype == torch.float32 else dict(atol=1.6e-2, rtol=1e-2) if dtype == torch.bfloat16 else dict(atol=2e-3, rtol=2e-3)
    torch.testing.assert_close(out.float(), ref.float(), **tol)      # (gi re-associated: a last-bit difference may cross a rounding boundary)
    torch.testing.assert_close(dots, rdots, atol=0.2 if dtype == torch.bfloat16 else 3e-2 if dtype == torch.float16 else 1e-4, rtol=1e-2)
    torch.testing.assert_close(dots.cpu(), out.float().cpu() @ gw[:, H:].t(), atol=3e-5, rtol=1e-5)   # dots are of the state AS STORED
    plain = ops.gru_gate_proj(cu(gh), cu(P), cu(b_ih), csr, cu(nd), cu(ed), cu(gb), cu(hp))
    assert torch.equal(plain, out)
    if dtype != torch.float32:
        # gh handed over in the state's 16-bit type (what the forward does since round 4): the same kernel arithmetic on the rounded values
        gh16 = cu(gh).to(dtype)
        a16 = ops.gru_gate_proj(gh16, cu(P), cu(b_ih), csr, cu(nd), cu(ed), cu(gb), cu(hp))
        b16 = ops.gru_gate_proj(gh16.float(), cu(P), cu(b_ih), csr, cu(nd), cu(ed), cu(gb), cu(hp))
        assert torch.equal(a16, b16)
        with pytest.raises(TypeError):
            ops.gru_gate_proj(gh16.to(torch.float16 if dtype == torch.bfloat16 else torch.bfloat16), cu(P), cu(b_ih), csr, cu(nd), cu(ed), cu(gb), cu(hp))


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_imp_ctx_at_chip_filling_size_matches_the_dense_formula(ops, dtype, monkeypatch):
    """128 images x (32 nodes, 992 edges) x H = 512 -- the launch `roofline_imp_large` times -- in the form the forward runs at that size
    (the sliced kernel) and in the persistent matrix-core form, against the dense formula (f32 torch on the GPU); then the routing
    threshold: from sgg_imp_ctx_mfma_min_units() (graph, 128-byte slice) units on (256 images) the matrix-core kernel runs."""
    from sgg_amd import _lib
    B, n, H = 128, 32, 512
    g = torch.Generator().manual_seed(9)

    def case(B):
        im = torch.arange(B).repeat_interleave(n)
        rel, _ = ops.pair_index_eval(cu(im))
        N, E = B * n, B * n * (n - 1)
        rel = rel[:E]
        e = cu(torch.randn(E, H, generator=g).to(dtype))
        nd, ed, gb = cu(torch.randn(N, 4, generator=g)), cu(torch.randn(E, 4, generator=g)), cu(torch.randn(4, generator=g))
        return rel, N, E, e, nd, ed, gb, ops.edge_csr(rel, N, cu(im), graphs=(B, n, n * (n - 1)))

    def forms(args, pair=2):
        rel, N, E, e, nd, ed, gb, csr = args
        out = {}
        for name, env in (('routed', None), ('mfma', 'm'), ('sliced', 's')):
            if env:
                monkeypatch.setenv('SGG_IMP_CTX', env)
            else:
                monkeypatch.delenv('SGG_IMP_CTX', raising=False)
            out[name] = ops.imp_ctx(e, csr, N, nd, ed, gb, pair=pair)
        monkeypatch.delenv('SGG_IMP_CTX', raising=False)
        return out
    args = case(B)
    rel, N, E, e, nd, ed, gb, csr = args
    s, o = rel[:, 1], rel[:, 2]
    tol = TOL16[dtype][1]
    for pair in (2, 0):
        ga = torch.sigmoid(nd[s, pair] + ed[:, pair] + gb[pair])
        gbv = torch.sigmoid(nd[o, pair + 1] + ed[:, pair + 1] + gb[pair + 1])
        exp_a, exp_b = _ctx_expect(e.float(), rel, ga, gbv, N)
        out = forms(args, pair)
        assert torch.equal(out['routed'], out['sliced']) and not torch.equal(out['mfma'], out['sliced'])   # (16-bit gates in the product: other bits)
        for got in (out['mfma'], out['sliced']):
            torch.testing.assert_close(got[0].float(), exp_a, **tol)
            torch.testing.assert_close(got[1].float(), exp_b, **tol)
        csum = ops.imp_ctx(e, csr, N, nd, ed, gb, pair=pair, ctx_sum=torch.empty((N, H), dtype=dtype, device=DEV))
        torch.testing.assert_close(csum.float(), exp_a + exp_b, **tol)
    del args, out, exp_a, exp_b
    units = _lib.load().sgg_imp_ctx_mfma_min_units()
    Bt = units // (H * 2 // 128)
    out = forms(case(Bt))
    assert torch.equal(out['routed'], out['mfma'])
    torch.testing.assert_close(out['mfma'].float(), out['sliced'].float(), **tol)
    out = forms(case(Bt - 1))
    assert torch.equal(out['routed'], out['sliced'])


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_imp_ctx_matrix_core_kernel_walks_several_units_per_workgroup(ops, dtype, monkeypatch):
    """The persistent kernel on ragged batches -- complete graphs of very different sizes, a sampled edge list with a node that lost its
    out-edges, a two-node graph -- with the grid capped so that every workgroup walks SEVERAL units (DMA ring across unit boundaries, the
    small arrays one unit ahead), small chunks (many per unit), both output modes: equal to the sliced kernel up to the 16-bit rounding of
    the gates, and bit-reproducible for every grid size.  A broken capacity promise poisons that graph's outputs in every form."""
    H = 128
    g = torch.Generator().manual_seed(77)
    for case, sizes in enumerate(([32, 5, 17, 2, 32, 9, 31, 3, 12], [32] * 12, [8, 2, 2])):
        im = np.concatenate([np.full(n, b) for b, n in enumerate(sizes)]).astype(np.int64)
        rel = O.get_rel_inds_eval(im)
        if case == 0:
            rng = np.random.RandomState(4)
            drop = (rel[:, 0] == 2) & (rng.rand(len(rel)) < 0.35)          # graph 2: sampled list ...
            drop |= (rel[:, 0] == 2) & (rel[:, 1] == im.tolist().index(2) + 4)   # ... and one node without out-edges
            rel = rel[~drop]
        N, E = len(im), len(rel)
        per_graph = [int((rel[:, 0] == b).sum()) for b in range(len(sizes))]
        e = cu(torch.randn(E, H, generator=g).to(dtype))
        nd, ed, gb = cu(torch.randn(N, 4, generator=g)), cu(torch.randn(E, 4, generator=g)), cu(torch.randn(4, generator=g))
        csr = ops.edge_csr(cu(rel), N, cu(im), graphs=(len(sizes), max(sizes), max(per_graph)))
        monkeypatch.setenv('SGG_IMP_CTX', 's')
        ref2 = ops.imp_ctx(e, csr, N, nd, ed, gb)
        refs = ops.imp_ctx(e, csr, N, nd, ed, gb, ctx_sum=torch.empty((N, H), dtype=dtype, device=DEV))
        assert torch.isfinite(ref2.float()).all()
        ctol = dict(atol=0.13, rtol=1.6e-2) if dtype == torch.bfloat16 else dict(atol=2e-2, rtol=2e-3)
        monkeypatch.setenv('SGG_IMP_CTX', 'm')
        for eb in ('', '64', '32'):
            if eb:
                monkeypatch.setenv('SGG_IMP_EB', eb)
            else:
                monkeypatch.delenv('SGG_IMP_EB', raising=False)
            first = None
            for cap in ('', '1', '3', '8', '11'):
                if cap:
                    monkeypatch.setenv('SGG_IMP_MAX_WGS', cap)
                else:
                    monkeypatch.delenv('SGG_IMP_MAX_WGS', raising=False)
                ctx2 = ops.imp_ctx(e, csr, N, nd, ed, gb)
                csum = ops.imp_ctx(e, csr, N, nd, ed, gb, ctx_sum=torch.empty((N, H), dtype=dtype, device=DEV))
                where = lambda m: '%s %s' % (m, (case, eb, cap))  # noqa: E731
                torch.testing.assert_close(ctx2.float(), ref2.float(), msg=where, **ctol)
                torch.testing.assert_close(csum.float(), refs.float(), msg=where, **ctol)
                if first is None:
                    first = (ctx2, csum)
                else:                                                    # the same form is bit-reproducible for every grid size
                    assert all(torch.equal(a, b) for a, b in zip(first, (ctx2, csum))), (case, eb, cap)
        for k_ in ('SGG_IMP_EB', 'SGG_IMP_MAX_WGS'):
            monkeypatch.delenv(k_, raising=False)
        # understated capacity (graph 0 has 56 edges, promised: 8): its outputs are NaN, the other graphs are untouched
        if case == 2:
            lie = ops.edge_csr(cu(rel), N, cu(im), graphs=(len(sizes), max(sizes), 8))
            for form in ('m', 's'):
                monkeypatch.setenv('SGG_IMP_CTX', form)
                monkeypatch.setenv('SGG_IMP_MAX_WGS', '2')
                ctx2 = ops.imp_ctx(e, lie, N, nd, ed, gb)
                assert torch.isnan(ctx2[:, :8].float()).all(), form
                torch.testing.assert_close(ctx2[:, 8:].float(), ref2[:, 8:].float(), **ctol)
    for k_ in ('SGG_IMP_CTX', 'SGG_IMP_EB', 'SGG_IMP_MAX_WGS'):
        monkeypatch.delenv(k_, raising=False)


def test_message_pass_with_and_without_the_graph_promise(ops):
    """Graphs the every-row-once kernels cannot hold (70 nodes > 64) walk the CSR lists; a batch they do take gives the same numbers with
    and without the promise; a promise that understates a graph poisons that graph's outputs instead of corrupting anything."""
    from sgg_amd.imp import ImpWeights, message_pass
    H = 64
    g = torch.Generator().manual_seed(5)
    sizes = [70, 3]                                  # 70 nodes > 64
    im = np.concatenate([np.full(n, b) for b, n in enumerate(sizes)]).astype(np.int64)
    rel = O.get_rel_inds_eval(im)
    N, E = len(im), len(rel)
    csr = ops.edge_csr(cu(rel), N, cu(im), graphs=(2, 70, 70 * 69))
    assert not ops.imp_sliced_ok(csr, H, torch.float32)
    v, e = cu(torch.randn(N, H, generator=g)), cu(torch.randn(E, H, generator=g))
    p = {}
    for gname in ('edge_gru', 'node_gru'):
        p[gname + '.weight_ih'], p[gname + '.weight_hh'] = torch.randn(3 * H, H, generator=g) / 8, torch.randn(3 * H, H, generator=g) / 8
        p[gname + '.bias_ih'], p[gname + '.bias_hh'] = torch.randn(3 * H, generator=g) / 8, torch.randn(3 * H, generator=g) / 8
    for gname in ('sub_vert_w_fc', 'obj_vert_w_fc', 'out_edge_w_fc', 'in_edge_w_fc'):
        p[gname + '.0.weight'], p[gname + '.0.bias'] = torch.randn(1, 2 * H, generator=g) / 8, torch.randn(1, generator=g)
    wts = ImpWeights.from_state({k: cu(t) for k, t in p.items()}, torch.float32)
    got = message_pass(e, v, cu(rel), csr, wts, 2, torch.float32)
    sd = {k: t for k, t in p.items()}
    exp_v, exp_e = O.message_pass(e.cpu(), v.cpu(), torch.from_numpy(rel[:, 1:]), sd, 2)
    torch.testing.assert_close(got[0].cpu(), exp_v, atol=3e-5, rtol=1e-5)
    torch.testing.assert_close(got[1].cpu(), exp_e, atol=3e-5, rtol=1e-5)
    # a batch the sliced kernel does take: same numbers with and without the hint
    sizes = [12, 9]
    im = np.concatenate([np.full(n, b) for b, n in enumerate(sizes)]).astype(np.int64)
    rel = O.get_rel_inds_eval(im)
    N, E = len(im), len(rel)
    v, e = cu(torch.randn(N, H, generator=g)), cu(torch.randn(E, H, generator=g))
    hint = ops.edge_csr(cu(rel), N, cu(im), graphs=(2, 12, 132))
    plain = ops.edge_csr(cu(rel), N, cu(im))
    assert ops.imp_sliced_ok(hint, H, torch.float32) and not ops.imp_sliced_ok(plain, H, torch.float32)
    a = message_pass(e, v, cu(rel), hint, wts, 3, torch.float32)
    b = message_pass(e, v, cu(rel), plain, wts, 3, torch.float32)
    torch.testing.assert_close(a[0], b[0], atol=2e-5, rtol=1e-5)
    torch.testing.assert_close(a[1], b[1], atol=2e-5, rtol=1e-5)
    # a hint that understates a graph (here: 132 edges promised as 100) must not corrupt anything silently: that graph's outputs are NaN
    lie = ops.edge_csr(cu(rel), N, cu(im), graphs=(2, 12, 100))
    ctx2 = ops.imp_ctx(e, lie, N, cu(torch.zeros(N, 4)), cu(torch.zeros(E, 4)), cu(torch.zeros(4)))
    assert torch.isnan(ctx2[:, :12]).all() and torch.isfinite(ctx2[:, 12:]).all()   # graph 0 has 12 nodes / 132 edges, graph 1 has 9 / 72


# ----------------------------------------------------------------------------------------- eval tail
def test_eval_tail_vs_reference_golden(ops, golden):
    g = golden('eval_tail')
    od, rd, rel = cu(g['obj_dists']), cu(g['rel_dists']), cu(g['rel_inds'])
    sc, pr, rels, ps = ops.eval_tail(od, rd, rel)
    np.testing.assert_array_equal(pr.cpu().numpy(), g['sg_classes'])
    np.testing.assert_allclose(sc.cpu().numpy(), g['sg_scores'], atol=1e-6)
    np.testing.assert_array_equal(rels.cpu().numpy(), g['sg_rels'])
    np.testing.assert_allclose(ps.cpu().numpy(), g['sg_pred_scores'], atol=1e-6)
    sc, pr, rels, ps = ops.eval_tail(od, rd, rel, cu(g['gt_classes']))
    np.testing.assert_array_equal(pr.cpu().numpy(), g['pc_classes'])
    np.testing.assert_array_equal(rels.cpu().numpy(), g['pc_rels'])
    np.testing.assert_allclose(ps.cpu().numpy(), g['pc_pred_scores'], atol=1e-6)


@pytest.mark.parametrize('E', [1, 992, 2450, 7936, 20000])
def test_eval_tail_sizes_sorted_and_permutation(ops, E):
    g = torch.Generator().manual_seed(E)
    N = 50
    od, rd = torch.randn(N, 151, generator=g), torch.randn(E, 51, generator=g)
    rel = torch.stack((torch.zeros(E, dtype=torch.long), torch.randint(0, N, (E,), generator=g),
                       torch.randint(0, N, (E,), generator=g)), 1)
    sc, pr, rels, ps = ops.eval_tail(cu(od), cu(rd), cu(rel))
    b, c, s, r, p = O.eval_tail(od, rd, rel, torch.zeros(N, 4))
    np.testing.assert_array_equal(pr.cpu().numpy(), c)
    score = ps.cpu()[:, 1:].max(1)[0] * sc.cpu()[rels.cpu()[:, 0]] * sc.cpu()[rels.cpu()[:, 1]]
    assert bool((score[:-1] >= score[1:] - 1e-7).all())          # sortedness
    np.testing.assert_allclose(np.sort(ps.cpu().numpy().sum(1)), np.sort(p.sum(1)), atol=1e-5)  # permutation of rows
    # same order as the stable oracle, except where two scores differ by less than fp32 rounding of the softmax
    ref_score = torch.from_numpy(p)[:, 1:].max(1)[0] * torch.from_numpy(s)[torch.from_numpy(r[:, 0])] * \
        torch.from_numpy(s)[torch.from_numpy(r[:, 1])]
    np.testing.assert_allclose(score.numpy(), ref_score.numpy(), rtol=1e-5, atol=1e-9)
    same = np.all(np.abs(ps.cpu().numpy() - p) < 1e-5, axis=1)
    assert same.mean() > 0.99
    for i in np.nonzero(~same)[0]:      # a mismatching row must sit in a near-tie
        gaps = [abs(float(ref_score[i]) - float(ref_score[j])) for j in (i - 1, i + 1) if 0 <= j < E]
        assert min(gaps) <= 2e-6 * float(ref_score[i])


@pytest.mark.parametrize('shape', [(64, 128, 128), (256, 256, 384), (7936, 512, 256), (1024, 1536, 512),
                                   (7936, 512, 4096), (512, 4096, 2048)])   # the last two: > 256 workgroups (two per CU)
@pytest.mark.parametrize('out_dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_gemm_tn_matches_transposed_product(ops, shape, out_dtype):
    """sgg_gemm_tn (reduction rows staged as they lie, ds_read_b64_tr_b16 fragments) == A^T . B, single pass and split-K."""
    Mred, N, K = shape
    g = torch.Generator().manual_seed(Mred + N)
    A = (torch.randn(Mred, N, generator=g) / 4).to(torch.bfloat16)
    B = torch.randn(Mred, K, generator=g).to(torch.bfloat16)
    ref = A.float().t() @ B.float()
    assert ops.gemm_tn_ok(cu(A), cu(B)) and not ops.gemm_tn_ok(cu(A[:, :100]), cu(B))
    for splits in (1, None, min(4, Mred // 64)):
        got = ops.gemm_tn(cu(A), cu(B), out_dtype=out_dtype, splits=splits)
        tol = dict(atol=2e-3 * (Mred ** 0.5), rtol=1e-4) if out_dtype == torch.float32 else dict(atol=0.02 * (Mred ** 0.5), rtol=2e-2)
        torch.testing.assert_close(got.float().cpu(), ref, **tol)
    # strided operands: column blocks of wider matrices
    wide = torch.randn(Mred, N + 128, generator=g).to(torch.bfloat16)
    got = ops.gemm_tn(cu(wide)[:, 128:], cu(B), out_dtype=torch.float32, splits=1)
    torch.testing.assert_close(got.cpu(), wide[:, 128:].float().t() @ B.float(), atol=2e-3 * (Mred ** 0.5), rtol=1e-4)


@pytest.mark.parametrize('shape', [(64, 4096, 2048), (192, 2048, 4096), (3968, 4096, 2048), (7936, 4096, 4096)])
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_gemm_tn_pingpong_form(ops, shape, dtype):
    """Whole 256x256 tiles, 128 of them or more: sgg_gemm_tn runs the ping-pong kernel's TN form (gemm256.hip: operands staged with the
    reduction index as the slow axis, fragments through ds_read_b64_tr_b16).  Against A^T . B in fp32 and against the 128x128 TN kernel
    (the split-K route); 2, 6, 124 and 248 K-tiles (prologue / peeled tail of the K loop); non-symmetric operands."""
    Mred, N, K = shape
    g = torch.Generator().manual_seed(Mred + K)
    A = (torch.randn(Mred, N, generator=g) / 4).to(dtype)
    B = (torch.randn(Mred, K, generator=g) + 0.25).to(dtype)
    ref = (cu(A).float().t() @ cu(B).float()).cpu()
    got = ops.gemm_tn(cu(A), cu(B), out_dtype=torch.float32, splits=1)
    torch.testing.assert_close(got.cpu(), ref, atol=2e-3 * (Mred ** 0.5), rtol=1e-4)
    if Mred >= 128:
        other = ops.gemm_tn(cu(A), cu(B), out_dtype=torch.float32, splits=2)        # 128x128 kernel, two K slices
        torch.testing.assert_close(got.cpu(), other.cpu(), atol=1e-3 * (Mred ** 0.5), rtol=1e-5)
    out16 = ops.gemm_tn(cu(A), cu(B), out_dtype=dtype, splits=1)
    torch.testing.assert_close(out16.float().cpu(), ref, atol=0.02 * (Mred ** 0.5), rtol=2e-2)
    # column blocks of wider matrices (row pitch != columns), as the fc6 weight gradient's main launch takes them
    wide = torch.randn(Mred, K + 256, generator=g).to(dtype)
    got = ops.gemm_tn(cu(A), cu(wide)[:, 256:], out_dtype=torch.float32, splits=1)
    torch.testing.assert_close(got.cpu(), (cu(A).float().t() @ cu(wide)[:, 256:].float()).cpu(), atol=2e-3 * (Mred ** 0.5), rtol=1e-4)


@pytest.mark.parametrize('Mred', [3973, 31, 64, 1000])
def test_gemm_tn256_any_row_count_and_group_addend(ops, Mred):
    """sgg_gemm_tn256: reduction rows that are no multiple of 32 pass through the zero-padded tail tile (the ragged batch of the bench
    has 3973 unordered pairs); the group addend of fc6's weight gradient in the epilogue; column blocks as operands."""
    g = torch.Generator().manual_seed(Mred)
    N, K, group = 2048, 4096 + 256, 49
    A = (torch.randn(Mred, N, generator=g) / 4).to(torch.float16)
    B = (torch.randn(Mred, K, generator=g) + 0.25).to(torch.float16)
    r = torch.randn(N, (K + group - 1) // group + 1, generator=g)
    ref = (cu(A).float().t() @ cu(B).float()).cpu()
    got = ops.gemm_tn256(cu(A), cu(B), out_dtype=torch.float32)
    torch.testing.assert_close(got.cpu(), ref, atol=2e-3 * (Mred ** 0.5), rtol=1e-4)
    add = r[:, torch.arange(K) // group]
    got = ops.gemm_tn256(cu(A), cu(B), out_dtype=torch.float32, gadd=(cu(r), group))
    torch.testing.assert_close(got.cpu(), ref + add, atol=2e-3 * (Mred ** 0.5), rtol=1e-4)
    got = ops.gemm_tn256(cu(A), cu(B)[:, 256:], out_dtype=torch.float32, gadd=(cu(r), group), col0=256)
    torch.testing.assert_close(got.cpu(), (ref + add)[:, 256:], atol=2e-3 * (Mred ** 0.5), rtol=1e-4)
    assert ops.gemm_tn_ok(cu(A), cu(B)) and ops.gemm_tn256_ok(cu(A), cu(B))
    torch.testing.assert_close(ops.gemm_tn(cu(A), cu(B)).cpu(), ref, atol=2e-3 * (Mred ** 0.5), rtol=1e-4)     # the routed entry


def test_gemm_tn_full_waves_and_pairsum(ops):
    """fc6's weight gradient as the backward runs it: pair sums of the edge rows (rows form == the transposed form), whole rounds on
    sgg_gemm_tn256 + the last tile columns on the 128 x 128 TN kernel's split form, group addend everywhere -- against one launch and
    against the transposes + NT route (gemm_full_waves)."""
    g = torch.Generator().manual_seed(5)
    U, E, N, K, group = 3968, 7936, 4096, 25088, 49
    d = cu((torch.randn(E, N, generator=g) / 8).to(torch.float16))
    ef = cu(torch.randn(U, K, generator=g).relu().to(torch.float16))
    u2e = torch.stack((torch.arange(U) * 2 + 1, torch.arange(U) * 2), 1).to(torch.int32)
    u2e[5, 1] = -1
    u2e[9] = -1
    u2e = cu(u2e)
    d6 = ops.pairsum(d, u2e)
    d6t = ops.transpose_pairsum(d, u2e)
    assert torch.equal(d6.t().contiguous(), d6t[:, :U].contiguous())
    r = cu(torch.randn(N, 512, generator=g))
    one = ops.gemm_tn256(d6, ef, out_dtype=torch.float16, gadd=(r, group))
    two = ops.gemm_tn_full_waves(d6, ef, out_dtype=torch.float16, gadd=(r, group))
    nt = ops.gemm_full_waves(d6t, ops.transpose(ef), out_dtype=torch.float16, gadd=(r, group))
    torch.testing.assert_close(two.float(), one.float(), atol=0.05, rtol=2e-3)
    torch.testing.assert_close(two.float(), nt.float(), atol=0.05, rtol=2e-3)
    assert torch.equal(two[:, :24576], one[:, :24576])          # the main launch computes these columns the same way


def test_gemm_full_waves_equals_one_launch():
    """the fc6 weight-gradient shape (16 x 98 tiles = 6.125 rounds): full-round launch + split-K tail vs the single launch, and a
    short reduction that must fall back to the single launch"""
    from sgg_amd import ops
    g = torch.Generator().manual_seed(0)
    for K in (7936, 128):
        A = (torch.randn(4096, K, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
        W = (torch.randn(25088, K, generator=g) * 0.1).to(torch.bfloat16).to(DEV)
        one = ops.gemm(A, W, out_dtype=torch.float32)
        two = ops.gemm_full_waves(A, W, out_dtype=torch.float32)
        assert two.shape == one.shape
        # same products; the tail columns are summed in two K halves instead of one pass (fp32 accumulation order)
        torch.testing.assert_close(two, one, atol=2e-3 * (K / 7936) ** 0.5 + 1e-5, rtol=1e-4)
        assert torch.equal(two[:, :24576], one[:, :24576])
