"""Stream-K scheduling of the 256x256 ping-pong launches (csrc/tile_sched.h): one persistent launch whose last
round's K-tiles are spread over all CUs.  A split tile is ONE accumulation chain handed from workgroup to workgroup, so every output
must equal the plain launch's BIT FOR BIT -- for every element type, epilogue, K segmentation and tile-count remainder, with the
published parts found (mode 1) or not (mode 3: every consumer recomputes its tile from K-tile 0), and over repeated launches (epochs)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture()
def ops():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from sgg_amd import ops
    prev = ops.streamk_mode(1)
    yield ops
    ops.streamk_mode(prev)
    ops.streamk_disable()


def both(ops, fn, modes=(1, 3)):
    """fn() with the plain launch, then under stream-K modes; returns (plain, [stream-K results])"""
    ops.streamk_disable()
    plain = fn()
    outs = []
    for m in modes:
        ops.streamk_mode(m)
        ops.streamk_enable()
        outs.append(fn())
    ops.streamk_mode(1)
    torch.cuda.synchronize()
    return plain, outs


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize('M,N,K', [(4608, 3840, 512), (4500, 4000, 1024), (7936, 4096, 2048), (2048, 33 * 256 + 40, 768)])
def test_gemm_streamk_is_bit_identical(ops, dtype, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(DEV).to(dtype)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV).to(dtype)
    bias = torch.randn(N, generator=g).to(DEV)
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    assert tiles > 256 and tiles % 256, tiles
    for kw in (dict(), dict(bias=bias, act=ops.ACT_RELU, out_dtype=torch.float32), dict(bias=bias, out_dtype=torch.bfloat16)):
        plain, outs = both(ops, lambda: ops.gemm(A, W, **kw))
        for o in outs:
            assert torch.equal(o, plain), (kw.keys(), float((o.float() - plain.float()).abs().max()))
    # a second K segment (fc6's fold: [A | A2] . [W | W2]^T)
    if dtype != torch.float32:
        A2 = torch.randn(M, 512, generator=g).to(DEV).to(dtype)
        W2 = (torch.randn(N, 512, generator=g) / 23).to(DEV).to(dtype)
        plain, outs = both(ops, lambda: ops.gemm(A, W, A2=A2, W2=W2, bias=bias))
        for o in outs:
            assert torch.equal(o, plain)
    # the reference product, loosely (the plain kernel itself is checked in test_kernels_gpu.py)
    ref = A.float() @ W.float().t()
    got = both(ops, lambda: ops.gemm(A, W, out_dtype=torch.float32), modes=(1,))[1][0]
    tol = 2e-3 if dtype == torch.float32 else 3e-2
    assert float((got - ref).abs().max()) <= tol * float(ref.abs().max())


def test_fc6_weight_gradient_shape_and_repeated_launches(ops):
    """[4096 x 3968] . [25088 x 3968]^T (1568 tiles = 6.125 rounds) with the group addend of the pair path, bf16 out: the shape the scheduler
    was built for; 12 launches in a row (epochs advance, slots are reused) while another stream keeps some CUs busy."""
    g = torch.Generator().manual_seed(1)
    U = 3968
    A = (torch.randn(4096, U, generator=g) / 50).to(DEV).to(torch.float16)
    W = torch.randn(25088, U, generator=g).relu().to(DEV).to(torch.float16)
    r = torch.randn(4096, 512, generator=g).to(DEV)
    ops.streamk_disable()
    plain = ops.gemm_full_waves(A, W, out_dtype=torch.bfloat16, gadd=(r, 49))      # main launch + split-K tail: NOT the single-chain order
    ops.streamk_enable()
    ops.streamk_mode(0)
    one = ops.gemm_full_waves(A, W, out_dtype=torch.bfloat16, gadd=(r, 49))         # mode 0 + workspace: one plain launch of all 1568 tiles
    ops.streamk_mode(1)
    side = torch.cuda.Stream()
    busy = torch.randn(4096, 4096, device=DEV)
    for it in range(12):
        if it % 3 == 0:
            with torch.cuda.stream(side):
                for _ in range(4):
                    busy = (busy @ busy).clamp_(-1, 1)
        out = ops.gemm_full_waves(A, W, out_dtype=torch.bfloat16, gadd=(r, 49))
        assert torch.equal(out, one), it
    torch.cuda.synchronize()
    # against round 3's two-launch form: equal up to the order of the fp32 sums in the last two tile columns
    assert torch.equal(out[:, :25088 - 512], plain[:, :25088 - 512])
    assert float((out.float() - plain.float()).abs().max()) <= 2e-2 * float(plain.float().abs().max())


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_implicit_gemm_conv_streamk_is_bit_identical(ops, dtype):
    """conv5-shaped layer (38 x 38 x 512 -> 512) on enough frames for 362 tiles: the CONV form of the same kernel"""
    B, H, C = 32, 38, 512
    g = torch.Generator().manual_seed(2)
    x = torch.zeros(B, H + 2, H + 2, C, dtype=dtype, device=DEV)
    x[:, 1:-1, 1:-1] = torch.randn(B, H, H, C, generator=g).to(DEV).to(dtype)
    w = (torch.randn(C, 3, 3, C, generator=g) / (9 * C) ** 0.5).to(DEV).to(dtype)
    b = torch.randn(C, generator=g).to(DEV)

    def run():
        out = torch.zeros(B, H + 2, H + 2, C, dtype=dtype, device=DEV)
        ops.conv3x3_relu(x, w, b, out, 1)
        return out
    plain, outs = both(ops, run)
    for o in outs:
        assert torch.equal(o, plain)
    ref = torch.nn.functional.conv2d(x[:, 1:-1, 1:-1].permute(0, 3, 1, 2).float(), w.permute(0, 3, 1, 2).float(), b, padding=1).relu()
    assert float((plain[:, 1:-1, 1:-1].permute(0, 3, 1, 2).float() - ref).abs().max()) <= 4e-2 * float(ref.abs().max())


def test_scheduler_plan_covers_every_k_tile_once():
    """host restatement of tile_sched.h's cut (sk_bound / sk_range): for many (tiles, K-tiles) every (tile, K-tile) unit of the stream-K
    region belongs to exactly one workgroup, no part is shorter than 8 K-tiles, no range holds a middle part"""
    def bound(i, G, units, nt):
        b = i * units // G
        t, off = divmod(b, nt)
        if off < 8:
            off = 0
        elif nt - off < 8:
            off, t = 0, t + 1
        else:
            off &= ~3
        return t * nt + off
    G = 256
    for tiles in (257, 300, 362, 400, 496, 511, 1568, 2049):
        for nt in (16, 31, 62, 124, 144, 784):
            R, rem = divmod(tiles, G)
            sk_tiles = G + rem
            units = sk_tiles * nt
            seen = [0] * units
            for lc in range(G):
                b0, b1 = bound(lc, G, units, nt), bound(lc + 1, G, units, nt)
                assert b1 - b0 >= nt - 3 and (b0 % nt == 0 or b0 // nt != (b1 - 1) // nt or b1 % nt == 0), (tiles, nt, lc)
                for part in ((b0, min(b1, (b0 // nt + 1) * nt)), (max(b0, (b1 // nt) * nt), b1)):
                    if part[1] > part[0] and (part[0] % nt or part[1] % nt):
                        assert part[1] - part[0] >= 8, (tiles, nt, lc, part)
                for u in range(b0, b1):
                    seen[u] += 1
            assert all(c == 1 for c in seen), (tiles, nt)
