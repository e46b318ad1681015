"""Host-side contract of sgg_amd/dense.py (SURVEY 8 f-4 b) without a GPU: parameter names / shapes of the layers that stand in for
nn.Conv2d / nn.Linear / spectral_norm / BatchNorm, the channels-last batch normalisation against torch's (it is plain torch arithmetic),
and that the contractions refuse CPU tensors instead of falling back."""
import pytest
import torch
import torch.nn as nn


def test_layers_carry_torch_parameter_names_and_refuse_cpu_tensors():
    from sgg_amd import dense
    conv = dense.Conv2d(6, 4, 3, padding=1)
    assert {k: tuple(v.shape) for k, v in conv.state_dict().items()} == {k: tuple(v.shape) for k, v in nn.Conv2d(6, 4, 3, padding=1).state_dict().items()}
    sn = dense.Conv2d(6, 4, 3, spectral=True)
    assert sorted(sn.state_dict()) == sorted(nn.utils.spectral_norm(nn.Conv2d(6, 4, 3)).state_dict())
    lin = dense.Linear(5, 3)
    assert isinstance(lin, nn.Linear) and sorted(lin.state_dict()) == ['bias', 'weight']
    with pytest.raises(RuntimeError):
        conv(torch.zeros(1, 5, 5, 6))
    with pytest.raises(RuntimeError):
        lin(torch.zeros(2, 5))


@pytest.mark.parametrize('shape', [(40, 12), (2, 5, 7, 12)])
def test_batchnorm_rows_is_torch_batchnorm_on_channels_last(shape):
    from sgg_amd import dense
    torch.manual_seed(0)
    C = shape[-1]
    mine = dense.BatchNormRows(C)
    ref = nn.BatchNorm1d(C) if len(shape) == 2 else nn.BatchNorm2d(C)
    with torch.no_grad():
        ref.weight.uniform_(0.5, 1.5), ref.bias.normal_()
    mine.load_state_dict(ref.state_dict())
    x = torch.randn(shape, requires_grad=True)
    xr = x if len(shape) == 2 else x.permute(0, 3, 1, 2)
    back = (lambda t: t) if len(shape) == 2 else (lambda t: t.permute(0, 2, 3, 1))
    a, r = mine(x), back(ref(xr))
    torch.testing.assert_close(a, r, atol=1e-5, rtol=1e-5)
    for k in ('running_mean', 'running_var', 'num_batches_tracked'):
        torch.testing.assert_close(mine.state_dict()[k].float(), ref.state_dict()[k].float(), atol=1e-6, rtol=1e-6)
    g = torch.randn_like(a)
    torch.testing.assert_close(torch.autograd.grad(a, x, g)[0], torch.autograd.grad(r, x, g)[0], atol=1e-5, rtol=1e-4)
    mine.eval(), ref.eval()
    torch.testing.assert_close(mine(x), back(ref(xr)), atol=1e-5, rtol=1e-5)


def test_sync_batchnorm_conversion_is_in_place_and_keeps_the_parameters():
    from sgg_amd import dense
    net = nn.Sequential(dense.Conv2d(4, 6, 1), dense.BatchNormRows(6), nn.ReLU(), nn.Sequential(dense.BatchNormRows(6)))
    w = net[1].weight
    assert dense.sync_batchnorm_(net) is net
    assert isinstance(net[1], dense.RowsSyncBatchNorm) and isinstance(net[3][0], dense.RowsSyncBatchNorm) and net[1].weight is w


def test_first_block_routing_is_switchable():
    """SGG_CONV1_FUSE / SGG_CONV_PP / lanes are read per call (tests and A/B runs flip them without reloading the library)"""
    import inspect
    from sgg_amd import detector, train
    assert "SGG_CONV1_FUSE" in inspect.getsource(detector.VGGDetector.features)
    src = inspect.getsource(train)
    assert "SGG_BWD_LANE" in src and "SGG_TRAIN_IMP_LANE" in src
