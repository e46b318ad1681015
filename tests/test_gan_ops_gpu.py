"""GAN generator data movement (started): boxes_to_layout forward / gradient and the GraphTripleConv gather / pooling
kernels against vectors produced by the reference's augment/layout.py and augment/graphconv.py (tests/golden/gan_ops.npz), against
the oracle at the GAN's sizes, and through adjoint identities.  fp32 tolerance 2e-5 (sum order), bf16 2e-2 relative."""
import os

import numpy as np
import pytest
import torch

from oracle import sgg_oracle as O

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'gan_ops.npz'))
DEV = 'cuda:0'


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t.to(dtype) if dtype is not None else t


@pytest.mark.parametrize('tag', ['patch', 'vec'])
@pytest.mark.parametrize('hw', [(38, 38), (10, 14)])
@pytest.mark.parametrize('pool', ['sum', 'avg'])
def test_boxes_to_layout_forward_and_gradient_equal_reference(tag, hw, pool):
    from sgg_amd.gan_ops import boxes_to_layout
    key = 'lay_%s_%dx%d_%s' % (tag, hw[0], hw[1], pool)
    v = _t(G['lay_%s_in' % tag]).requires_grad_(True)
    out = boxes_to_layout(v, _t(G['lay_boxes']), _t(G['lay_img']), hw[0], hw[1], pooling=pool)
    assert tuple(out.shape) == G[key + '_out'].shape
    np.testing.assert_allclose(out.detach().cpu().numpy(), G[key + '_out'], atol=2e-5)
    (out * _t(G[key + '_up'])).sum().backward()
    np.testing.assert_allclose(v.grad.cpu().numpy(), G[key + '_din'], atol=5e-5)


def test_boxes_to_layout_rejects_unknown_pooling():
    from sgg_amd.gan_ops import boxes_to_layout
    with pytest.raises(ValueError):
        boxes_to_layout(_t(G['lay_vec_in']), _t(G['lay_boxes']), _t(G['lay_img']), 8, pooling='max')


def test_boxes_to_layout_gan_size_vs_oracle_and_bf16():
    """the GAN's configuration: 8 images x 32 objects, 512 channels, 7x7 patches -> 38x38 (augment/gan.py:202-207)"""
    from sgg_amd.gan_ops import boxes_to_layout_nhwc
    rng = np.random.RandomState(5)
    B, nb, D, S, Hh = 8, 32, 512, 7, 38
    xy = rng.uniform(0, 0.7, size=(B * nb, 2)); wh = rng.uniform(0.03, 0.4, size=(B * nb, 2))
    boxes = np.concatenate((xy, np.minimum(xy + wh, 1.0)), 1).astype(np.float32)
    img = np.repeat(np.arange(B), nb).astype(np.int64)
    v = rng.randn(B * nb, S, S, D).astype(np.float32)
    want = O.boxes_to_layout(v.transpose(0, 3, 1, 2), boxes, img, Hh, Hh, 'sum').transpose(0, 2, 3, 1)
    got = boxes_to_layout_nhwc(_t(v), _t(boxes), _t(img), Hh, Hh, 'sum')
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=1e-4, rtol=1e-5)      # up to 32 overlapping objects summed in fp32, different order
    got16 = boxes_to_layout_nhwc(_t(v, torch.bfloat16), _t(boxes), _t(img), Hh, Hh, 'sum').float().cpu().numpy()
    assert np.abs(got16 - want).max() <= 2e-2 * np.abs(want).max()
    # adjoint identity <L v, u> == <v, L^T u> at full size (the gradient kernel is the transpose of the forward)
    vt = _t(v).requires_grad_(True)
    u = torch.randn(B, Hh, Hh, D, device=DEV)
    out = boxes_to_layout_nhwc(vt, _t(boxes), _t(img), Hh, Hh, 'sum')
    (out * u).sum().backward()
    v2 = torch.randn_like(vt)
    lhs = float((boxes_to_layout_nhwc(v2, _t(boxes), _t(img), Hh, Hh, 'sum') * u).double().sum())
    rhs = float((v2 * vt.grad).double().sum())
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0)


@pytest.mark.parametrize('k', [0, 1, 2])
def test_graph_triple_conv_layer_equals_reference(k):
    """one scene-graph convolution: HIP gather -> Linear/ReLU/Linear (library GEMMs) -> HIP pooling -> Linear/ReLU/Linear, with the
    reference layer's weights; outputs vs the reference layer's, gradients vs torch autograd of the same layer in plain torch."""
    from sgg_amd.gan_ops import triple_gather, triple_pool
    final, avg, dout = [int(v) for v in G['gc%d_cfg' % k]]
    Hd = int(G['gc_hidden'])
    w = {n[len('gc%d_' % k):]: _t(G[n]) for n in G.files if n.startswith('gc%d_net' % k)}
    lin = lambda x, a: torch.nn.functional.linear(x, w[a + '.weight'], w[a + '.bias'])
    edges = _t(G['gc_edges'])
    O_ = G['gc_obj'].shape[0]

    def layer(obj, pred, gather, pool):
        t = lin(torch.relu(lin(gather(obj, pred), 'net1.0')), 'net1.2')
        if final:
            t = torch.relu(t)
        else:                                              # graphconv.py:86-88: only the subject / object parts
            t = torch.cat((torch.relu(t[:, :Hd]), t[:, Hd:Hd + dout], torch.relu(t[:, Hd + dout:])), 1)
        h = lin(torch.relu(lin(pool(t), 'net2.0')), 'net2.2')
        return (torch.relu(h) if final else h), t[:, Hd:Hd + dout]

    def ref_gather(obj, pred):
        return torch.cat((obj[edges[:, 0]], pred, obj[edges[:, 1]]), 1)

    def ref_pool(t):
        out = torch.zeros(O_, Hd, device=DEV).index_add(0, edges[:, 0], t[:, :Hd]).index_add(0, edges[:, 1], t[:, Hd + dout:])
        if avg:
            cnt = torch.bincount(edges.reshape(-1), minlength=O_).clamp(min=1).float()
            out = out / cnt[:, None]
        return out

    obj, pred = _t(G['gc_obj']).requires_grad_(True), _t(G['gc_pred']).requires_grad_(True)
    no, npred = layer(obj, pred, lambda o, p: triple_gather(o, p, edges),
                      lambda t: triple_pool(t, edges, O_, Hd, Hd + dout, 'avg' if avg else 'sum'))
    np.testing.assert_allclose(no.detach().cpu().numpy(), G['gc%d_out_obj' % k], atol=2e-5)
    np.testing.assert_allclose(npred.detach().cpu().numpy(), G['gc%d_out_pred' % k], atol=2e-5)
    g = torch.Generator().manual_seed(k)
    u1, u2 = torch.randn(no.shape, generator=g).to(DEV), torch.randn(npred.shape, generator=g).to(DEV)
    ((no * u1).sum() + (npred * u2).sum()).backward()
    obj2, pred2 = _t(G['gc_obj']).requires_grad_(True), _t(G['gc_pred']).requires_grad_(True)
    ro, rp = layer(obj2, pred2, ref_gather, ref_pool)
    ((ro * u1).sum() + (rp * u2).sum()).backward()
    torch.testing.assert_close(obj.grad, obj2.grad, atol=2e-5, rtol=1e-4)
    torch.testing.assert_close(pred.grad, pred2.grad, atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize('tag,norm', [('net', 'none'), ('netbn', 'batch')])
def test_graph_triple_conv_net_loads_reference_weights_and_matches(tag, norm):
    """GraphTripleConvNet as augment/gan.py:109-115 builds it: the reference's state_dict loads by name (strict) and the outputs agree;
    'batch' = BatchNorm1d in train mode (batch statistics), as the reference module was run."""
    from sgg_amd.gan_ops import GraphTripleConvNet
    net = GraphTripleConvNet(G['gc_obj'].shape[1], input_edge_dim=G['gc_pred'].shape[1], output_dim=20, num_layers=3,
                             hidden_dim=int(G['gc_hidden']), pooling='avg', mlp_normalization=norm).to(DEV)
    sd = {n[len(tag) + 1:]: torch.from_numpy(G[n]) for n in G.files if n.startswith(tag + '_gconvs')}
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all('num_batches' in m for m in missing), (missing, unexpected)
    no, npred = net(_t(G['gc_obj']), _t(G['gc_pred']), _t(G['gc_edges']))
    np.testing.assert_allclose(no.detach().cpu().numpy(), G[tag + '_out_obj'], atol=5e-5, rtol=1e-4)
    np.testing.assert_allclose(npred.detach().cpu().numpy(), G[tag + '_out_pred'], atol=5e-5, rtol=1e-4)
    (no.sum() + npred.sum()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
