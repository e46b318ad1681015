"""Run-to-run reproducibility at BENCHMARK grid sizes.

Kernels whose workgroups share a CU (two per CU at 592x592) can hide races that the small-shape unit tests -- one workgroup per
CU -- never see.  Round 2 found one this way: conv1_2 (bf16, fused pool) produced a few hundred wrong outputs in 24 % of its
launches, because a bare s_barrier let LDS reads that had been issued but not completed race against the refill of their buffer
(common.h: lds_reads_done_barrier).  A kernel without atomics must give the same bits every time; these tests launch the
full-size shapes repeatedly and compare bitwise, and check the first result against an independent fp32 computation."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def model():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import sgg_amd
    from sgg_amd.synthetic import SyntheticData, init_weights
    return init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(DEV).eval()


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16, torch.float32])
def test_every_vgg_layer_is_bit_reproducible_at_608(model, dtype):
    from sgg_amd import ops
    from sgg_amd.detector import VGG16_CFG
    B, reps = 4, 25 if dtype != torch.float32 else 6
    ws = model.detector.prepared(dtype)
    g = torch.Generator().manual_seed(0)
    H = W = 608
    x = torch.zeros((B, H + 2, W + 2, 4), dtype=torch.float32, device=DEV)
    x[:, 1:-1, 1:-1, :3] = torch.randn(B, H, W, 3, generator=g).to(DEV)
    ci_layer, li, cfg = 0, 0, list(VGG16_CFG)
    while li < len(cfg):
        if cfg[li] == 'M':
            y = torch.zeros((B, H // 2 + 2, W // 2 + 2, x.shape[3]), dtype=dtype, device=DEV)
            ops.maxpool2x2(x, y, 1)
            H, W = H // 2, W // 2
        else:
            w, bias, ci, co = ws[ci_layer][:4]
            last = ci_layer == len(ws) - 1
            fuse = ci_layer > 0 and li + 1 < len(cfg) and cfg[li + 1] == 'M' and ops.conv_pool_fusable(H, W, co)
            if fuse:
                shape = (B, H // 2 + 2, W // 2 + 2, co)
                run = lambda out, x=x, w=w, bias=bias, ci=ci, co=co: ops.conv3x3_relu(x, w.view(co, 3, 3, ci), bias, out, 1, pool=True)
            elif ci_layer == 0:
                shape = (B, H + 2, W + 2, co)
                run = lambda out, x=x, w=w, bias=bias: ops.conv1_1(x, w, bias, out)
            else:
                shape = (B, H, W, co) if last else (B, H + 2, W + 2, co)
                run = lambda out, x=x, w=w, bias=bias, ci=ci, co=co, last=last: ops.conv3x3_relu(x, w.view(co, 3, 3, ci), bias, out, 0 if last else 1)
            y = torch.zeros(shape, dtype=dtype, device=DEV)
            run(y)
            if ci_layer > 0:                          # first result against torch's own convolution in fp32
                xi = x[:, 1:-1, 1:-1].permute(0, 3, 1, 2).float()
                ref = torch.nn.functional.conv2d(xi, w.view(co, 3, 3, ci).permute(0, 3, 1, 2).float(), bias, padding=1).relu()
                if fuse:
                    ref = torch.nn.functional.max_pool2d(ref, 2)
                got = (y if last else y[:, 1:-1, 1:-1]).float().permute(0, 3, 1, 2)
                tol = (2e-2 * ref.abs() + 6e-2) if dtype != torch.float32 else (1e-4 * ref.abs() + 1e-3)
                assert int(((got - ref).abs() > tol).sum()) == 0, 'conv layer %d: outputs off' % ci_layer
            for r in range(reps):
                y2 = torch.zeros(shape, dtype=dtype, device=DEV)
                run(y2)
                assert torch.equal(y, y2), 'conv layer %d (Cin %d, Cout %d%s): run %d differs from the first' % (
                    ci_layer, ci, co, ', fused pool' if fuse else '', r)
            if fuse:
                H, W = H // 2, W // 2
                li += 1
            ci_layer += 1
        x = y
        li += 1


def test_tile_gemm_is_bit_reproducible_with_two_workgroups_per_cu():
    """The 128x128 tile kernel at shapes of the step (many tiles, 2 workgroups per CU): same bits every launch, bf16 and f32."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from sgg_amd import ops
    g = torch.Generator().manual_seed(2)
    for dtype, (M, N, K) in ((torch.bfloat16, (7936, 1536, 512)), (torch.bfloat16, (7936, 512, 4096)), (torch.float32, (2048, 512, 1024))):
        A = torch.randn(M, K, generator=g).to(DEV).to(dtype)
        Wt = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV).to(dtype)
        b = torch.randn(N, generator=g).to(DEV)
        first = ops.gemm(A, Wt, b, ops.ACT_RELU)
        ref = (A.float() @ Wt.float().t() + b).relu()
        tol = 3e-2 if dtype == torch.bfloat16 else 1e-3
        assert float((first.float() - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))
        for r in range(20):
            assert torch.equal(first, ops.gemm(A, Wt, b, ops.ACT_RELU)), (dtype, M, N, K, r)


def test_detector_feature_map_and_eval_forward_are_bit_reproducible(model):
    from sgg_amd.synthetic import synthetic_batch
    batch = synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=111)
    images = [batch[0][i] for i in range(8)]
    model.set_compute_dtype(torch.bfloat16)
    with torch.no_grad():
        f0 = model.detector.features(images, torch.bfloat16)[0].clone()
        out0 = model([batch])
        for r in range(10):
            f = model.detector.features(images, torch.bfloat16)[0]
            assert torch.equal(f0, f), 'feature map: run %d differs' % r
        for r in range(3):
            out = model([batch])
            for a, b in zip(out0, out):
                np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_fused_first_block_is_bit_reproducible_and_equals_the_two_launches_at_608(model, dtype):
    """sgg_conv1_block at the detector's size (two workgroups share a CU's LDS: the producer's patch writes, the weight DMA and the
    fragment reads of the neighbour all interleave) -- 25 launches, identical bits, equal to conv1_1 followed by conv1_2 + pool."""
    from sgg_amd import ops
    ws = model.detector.prepared(dtype)
    B, H, W = 4, 608, 608
    g = torch.Generator().manual_seed(1)
    x = torch.zeros((B, H + 2, W + 2, 4), dtype=torch.float32, device=DEV)
    x[:, 1:-1, 1:-1, :3] = torch.randn(B, H, W, 3, generator=g).to(DEV)
    (w1, b1), (w2, b2) = ws[0][:2], ws[1][:2]
    y1 = torch.zeros((B, H + 2, W + 2, 64), dtype=dtype, device=DEV)
    ops.conv1_1(x, w1, b1, y1)
    ref = torch.zeros((B, H // 2 + 2, W // 2 + 2, 64), dtype=dtype, device=DEV)
    ops.conv3x3_relu(y1, w2.view(64, 3, 3, 64), b2, ref, 1, pool=True)
    frags = ws[0][4]
    for r in range(25):
        y = torch.zeros_like(ref)
        ops.conv1_block(x, frags, b1, w2.view(64, 3, 3, 64), b2, y, 1, pool=True)
        assert torch.equal(y, ref), 'run %d differs' % r
