"""sgg_amd/dense.py (SURVEY 8 f-4 b: the GAN's Linear / Conv2d layers on this package's GEMM) against torch's fp32 layers: forward,
input / weight / bias gradients, spectral normalisation state, channels-last batch normalisation."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def close(a, b, tol=2e-5):
    scale = max(1.0, float(b.abs().max()))
    assert float((a - b).abs().max()) <= tol * scale, (float((a - b).abs().max()), scale)


@pytest.mark.parametrize('M,K,N', [(37, 29, 13), (256, 216, 64), (1, 8, 1), (700, 663, 5)])
def test_affine_forward_and_gradients(M, K, N):
    from sgg_amd import dense
    torch.manual_seed(M + K)
    x = torch.randn(M, K, device=DEV, requires_grad=True)
    w = (torch.randn(N, K, device=DEV) / K ** 0.5).requires_grad_()
    b = torch.randn(N, device=DEV, requires_grad=True)
    g = torch.randn(M, N, device=DEV)
    y = dense.affine(x, w, b)
    ref = F.linear(x, w, b)
    close(y, ref)
    mine = torch.autograd.grad(y, (x, w, b), g)
    want = torch.autograd.grad(ref, (x, w, b), g)
    for a, r in zip(mine, want):
        close(a, r)
    # no bias, leading batch axes
    y3 = dense.affine(x.view(1, M, K), w)
    close(y3, F.linear(x, w).view(1, M, N))


@pytest.mark.parametrize('B,H,C,Cout,k,pad', [(3, 7, 16, 32, 3, 1), (2, 7, 45, 24, 3, 0), (2, 38, 8, 16, 3, 1), (5, 3, 12, 1, 3, 0),
                                              (4, 7, 48, 16, 1, 0), (1, 9, 5, 7, 5, 2)])
def test_conv2d_forward_and_gradients(B, H, C, Cout, k, pad):
    from sgg_amd import dense
    torch.manual_seed(H * C)
    x = torch.randn(B, H, H, C, device=DEV, requires_grad=True)             # channels-last
    w = (torch.randn(Cout, C, k, k, device=DEV) / (C * k * k) ** 0.5).requires_grad_()
    b = torch.randn(Cout, device=DEV, requires_grad=True)
    y = dense.conv2d(x, w, b, padding=pad)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=pad).permute(0, 2, 3, 1)
    assert y.shape == ref.shape
    close(y, ref)
    g = torch.randn_like(ref)
    mine = torch.autograd.grad(y, (x, w, b), g)
    want = torch.autograd.grad(ref, (x, w, b), g)
    for a, r in zip(mine, want):
        close(a, r, 5e-5)


def test_empty_batches_pass_through():
    from sgg_amd import dense
    x = torch.zeros(0, 7, 7, 8, device=DEV, requires_grad=True)
    w = torch.randn(4, 8, 3, 3, device=DEV, requires_grad=True)
    y = dense.conv2d(x, w, None, padding=1)
    assert y.shape == (0, 7, 7, 4)
    y.sum().backward()
    assert w.grad.abs().max() == 0 and x.grad.shape == x.shape


def test_cpu_tensors_are_refused():
    from sgg_amd import dense
    with pytest.raises(RuntimeError):
        dense.affine(torch.zeros(4, 8), torch.zeros(2, 8))


def test_spectral_conv_follows_torch_spectral_norm_state_and_values():
    from sgg_amd import dense
    torch.manual_seed(5)
    ref = nn.utils.spectral_norm(nn.Conv2d(12, 9, 3)).to(DEV)
    mine = dense.Conv2d(12, 9, 3, spectral=True).to(DEV)
    assert sorted(mine.state_dict()) == sorted(ref.state_dict())
    mine.load_state_dict(ref.state_dict())
    x = torch.randn(4, 7, 7, 12, device=DEV)
    for step in range(3):                       # training: one power-iteration step per call, state advances
        a = mine(x)
        r = ref(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
        close(a, r, 1e-4)
        close(mine.weight_u, ref.weight_u, 1e-5)
        close(mine.weight_v, ref.weight_v, 1e-5)
    ga = torch.autograd.grad(mine(x).square().sum(), mine.weight_orig)[0]
    gr = torch.autograd.grad(ref(x.permute(0, 3, 1, 2)).square().sum(), ref.weight_orig)[0]
    close(ga, gr, 2e-4)
    mine.eval(), ref.eval()                     # evaluation: the stored vectors, no iteration
    u0 = mine.weight_u.clone()
    close(mine(x), ref(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1), 1e-4)
    assert torch.equal(mine.weight_u, u0)


@pytest.mark.parametrize('shape', [(50, 24), (3, 5, 5, 24)])
def test_batchnorm_rows_equals_torch_batchnorm(shape):
    from sgg_amd import dense
    torch.manual_seed(2)
    C = shape[-1]
    mine = dense.BatchNormRows(C).to(DEV)
    ref = (nn.BatchNorm1d(C) if len(shape) == 2 else nn.BatchNorm2d(C)).to(DEV)
    assert sorted(mine.state_dict()) == sorted(ref.state_dict())
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5), ref.bias.normal_()
    mine.load_state_dict(ref.state_dict())
    x = torch.randn(shape, device=DEV, requires_grad=True)
    xr = x if len(shape) == 2 else x.permute(0, 3, 1, 2)
    back = (lambda t: t) if len(shape) == 2 else (lambda t: t.permute(0, 2, 3, 1))
    for _ in range(2):
        a, r = mine(x), back(ref(xr))
        close(a, r, 1e-5)
    for k in ('running_mean', 'running_var', 'num_batches_tracked'):
        close(mine.state_dict()[k].float(), ref.state_dict()[k].float(), 1e-6)
    g = torch.randn_like(a)
    close(torch.autograd.grad(a, x, g)[0], torch.autograd.grad(r, x, g)[0], 1e-4)
    mine.eval(), ref.eval()
    close(mine(x), back(ref(xr)), 1e-5)


def test_sync_batchnorm_conversion_keeps_parameters_and_layout():
    from sgg_amd import dense
    net = nn.Sequential(dense.Conv2d(4, 6, 1), dense.BatchNormRows(6), nn.ReLU())
    w = net[1].weight
    dense.sync_batchnorm_(net)
    assert isinstance(net[1], dense.RowsSyncBatchNorm) and net[1].weight is w
