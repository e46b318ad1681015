"""BASELINE configs[4] with GQA's vocabulary (missing 2 / next 4): `backbone='resnet50'` (config.py:76-78 forces it for GQA),
1 704 object and 311 predicate classes through the heads, the eval tail, the cross-entropy losses, the per-class NMS of SGDet and the
class-conditioned discriminators of the GAN (augment/gan.py:222-231: the one-hot class planes are 1 704 / 311 channels wide) -- at a
reduced frame size (256 px: the oracle's ResNet-50-FPN forward finishes in seconds), against the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import sgg_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
S = 256
NOBJ, NPRED = 1704, 311


def _model(mode, S=S):
    import sgg_amd
    from sgg_amd.resnet_fpn import FrozenBatchNorm2d
    from sgg_amd.synthetic import GQASyntheticData, init_weights
    torch.manual_seed(5)
    model = init_weights(sgg_amd.RelModelStanford(GQASyntheticData(), mode=mode, backbone='resnet50', min_size=S, max_size=S))
    for m in model.modules():
        if isinstance(m, FrozenBatchNorm2d):
            m.weight.uniform_(0.5, 1.0)
            m.running_var.uniform_(0.6, 1.4)
            m.running_mean.normal_(0, 0.1)
    return model


def _batch(B=2, n_boxes=6, n_fg=4, seed=21, S=S):
    from sgg_amd.synthetic import relabel_batch, synthetic_batch
    return relabel_batch(synthetic_batch(B=B, S=S, n_boxes=n_boxes, n_fg=n_fg, seed=seed), NOBJ, NPRED, seed=seed)


@pytest.fixture(scope='module')
def sgcls():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    model = _model('sgcls')
    sd = {k: v.detach().float().clone() for k, v in model.state_dict().items()}
    assert tuple(sd['obj_fc.weight'].shape) == (NOBJ, 512) and tuple(sd['rel_fc.weight'].shape) == (NPRED, 512)
    assert tuple(sd['detector.roi_heads.box_predictor.cls_score.weight'].shape) == (NOBJ, 1024)
    assert tuple(sd['detector.roi_heads.box_predictor.bbox_pred.weight'].shape) == (4 * NOBJ, 1024)
    model.to(DEV).eval().set_compute_dtype(torch.float32)
    batch = _batch()
    assert int(batch[4][:, 1].max()) > 150 and int(batch[5][:, 3].max()) > 50          # labels beyond VG's vocabulary are in the batch
    return model, sd, batch


def test_eval_forward_with_the_gqa_vocabulary_matches_oracle(sgcls):
    model, sd, batch = sgcls
    dev_batch = tuple(t_.to(DEV) if isinstance(t_, torch.Tensor) else t_ for t_ in batch)
    with torch.no_grad():
        boxes, cls, scores, rels, pred = model([dev_batch])
    ref = O.forward_gtbox(batch[0], batch[3].numpy(), batch[4].numpy(), batch[5].numpy(), sd, mode='sgcls', min_size=S, max_size=S)
    rb, rc, rs, rr, rp = ref['dets']
    assert pred.shape == (len(rr), NPRED) and ref['rm_obj_dists'].shape[1] == NOBJ
    np.testing.assert_array_equal(cls, rc)
    assert int(cls.max()) > 150                                        # the argmax really lands beyond VG's 151 classes
    np.testing.assert_allclose(scores, rs, atol=1e-3)
    key = lambda r: r[:, 0] * 100000 + r[:, 1]       # noqa: E731
    go, ro = np.argsort(key(rels)), np.argsort(key(rr))
    np.testing.assert_array_equal(rels[go], rr[ro])
    np.testing.assert_allclose(pred[go], rp[ro], atol=1e-3)
    assert (pred[:, 1:].argmax(1) + 1).max() > 50
    # the 16-bit default mode: same structure, scores within the mode's tolerance
    model.set_compute_dtype(torch.float16)
    try:
        with torch.no_grad():
            b2, c2, s2, r2, p2 = model([dev_batch])
    finally:
        model.set_compute_dtype(torch.float32)
    assert p2.shape == pred.shape and np.isfinite(p2).all() and np.allclose(p2.sum(1), 1.0, atol=2e-2)
    g2 = np.argsort(key(r2))
    np.testing.assert_array_equal(r2[g2], rr[ro])
    np.testing.assert_allclose(p2[g2], rp[ro], atol=3e-2)


def test_train_step_with_the_gqa_vocabulary_matches_oracle_autograd(sgcls):
    """train-mode logits on [N, 1704] / [E, 311], the fused cross-entropy (node + edge losses of lib/losses.py) against the oracle's, every
    head gradient against torch autograd of the oracle, then Trainer steps in f16 lower the loss"""
    from sgg_amd.train import param_names
    from sgg_amd.trainer import Trainer
    model, sd, batch = sgcls
    dev_batch = tuple(t_.to(DEV) if isinstance(t_, torch.Tensor) else t_ for t_ in batch)
    model.train()
    model.dropout_p = 0.0
    try:
        res = model([dev_batch])
        assert res.rm_obj_dists.shape[1] == NOBJ and res.rel_dists.shape[1] == NPRED
        tr = Trainer(model, lr=1e-3, pipeline=False, graph=False)
        loss = tr.losses(res)
        model.zero_grad()
        loss.backward()
        names = param_names(model)
        pq = {k: v.clone().requires_grad_(k in set(names)) for k, v in sd.items()}
        od, rd = O.predict(res.node_feat.float().cpu().contiguous(), res.edge_feat.float().cpu().contiguous(), res.rel_inds.cpu().numpy(),
                           res.rois.cpu().numpy(), pq, training=True)
        torch.testing.assert_close(res.rm_obj_dists.detach().cpu(), od.detach(), atol=1e-3, rtol=1e-3)
        torch.testing.assert_close(res.rel_dists.detach().cpu(), rd.detach(), atol=1e-3, rtol=1e-3)
        ref_loss = O.node_losses(od, res.rm_obj_labels.cpu()) + O.edge_losses(rd, res.rel_labels[:, -1].cpu(), 'baseline')
        assert abs(float(loss) - float(ref_loss)) < 1e-3 * max(1.0, abs(float(ref_loss))), (float(loss), float(ref_loss))
        ref_loss.backward()
        named = dict(model.named_parameters())
        for n in names:
            ref, got = pq[n].grad, named[n].grad.float().cpu()
            err = float((got - ref).abs().max()) / (float(ref.abs().max()) + 1e-6)
            assert got.shape == ref.shape and err < 2e-3, (n, err)
        model.set_compute_dtype(torch.float16)
        tr = Trainer(model, lr=1e-3)
        losses = [float(tr.step(dev_batch)) for _ in range(5)]
        tr.flush()
        assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    finally:
        model.load_state_dict(sd)
        model.to(DEV).eval().set_compute_dtype(torch.float32)


def test_sgdet_per_class_nms_over_1703_classes_matches_oracle():
    """SGDet with the GQA detector: 1 703 x proposals candidates per image into the per-class NMS (csrc/det.hip), the detection sets
    equal to the oracle's (boxes, labels, scores one to one)"""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from sgg_amd.synthetic import spread_detector_
    from tests.test_sgdet_gpu import _assert_same_detections
    model = _model('sgdet')
    sd = spread_detector_({k: v.detach().float().clone() for k, v in model.state_dict().items()}, grow=3.0)
    model.load_state_dict(sd)
    model.to(DEV).eval().set_compute_dtype(torch.float32)
    batch = _batch(B=2, n_boxes=4, n_fg=2, seed=9)
    for thresh in (0.05, 0.3):
        model.set_box_score_thresh(thresh)
        with torch.no_grad():
            ref = O.sgdet_detect(batch[0], sd, score_thresh=thresh, min_size=S, max_size=S, backbone='resnet50')
            res = model.faster_rcnn(batch[0], None, batch[4].to(DEV), None)
        n = _assert_same_detections(res, ref[6])
        assert n >= 4, n
        assert int(res.rm_obj_labels.max()) > 150 and int(res.rm_obj_labels.max()) < NOBJ


def test_gan_iteration_with_the_gqa_vocabulary():
    """main.py:124-194 on the resnet50 / GQA model: the class-conditioned discriminators see 256 + 1704 and 256 + 311 input channels
    (augment/gan.py:222-231), the generator embeds 1 704 / 311 classes; one iteration gives the seven losses, finite, moves G, D and the
    SGG head; the discriminator's first layer on the one-hot planes against a torch fp32 convolution"""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import torch.nn.functional as F
    from sgg_amd.feature_gan import GAN, gan_train_step
    from sgg_amd.synthetic import GQASyntheticData
    FS = 1333                     # the config's frame size: 21 x 21 'pool'-level maps, what the global discriminator's layer table is built for
    model = _model('sgcls', S=None).to(DEV)
    assert model.detector.transform.min_size == FS
    model.set_compute_dtype(torch.float32)
    model.dropout_p = 0.0
    for n, p in model.named_parameters():
        if n.startswith('detector.'):
            p.requires_grad = False
    data = GQASyntheticData()
    torch.manual_seed(11)
    gan = GAN(data.ind_to_classes, data.ind_to_predicates, n_ch=model.edge_dim, pool_sz=model.pool_sz, fmap_sz=model.fmap_sz, n_layers_G=2,
              device=DEV).to(DEV)
    assert gan.D_nodes[0].weight_orig.shape[1] == 256 + NOBJ and gan.D_edges[0].weight_orig.shape[1] == 256 + NPRED
    assert tuple(gan.G_obj_embed.weight.shape) == (NOBJ, 200) and tuple(gan.G_rel_embed.weight.shape) == (NPRED, 200)
    batch = _batch(B=2, n_boxes=5, n_fg=4, seed=31, S=FS)
    dev_batch = tuple(t_.to(DEV) if isinstance(t_, torch.Tensor) else t_ for t_ in batch)
    dev_batch = ([im.to(DEV) for im in batch[0]],) + dev_batch[1:]
    model.train()
    res = model([dev_batch])
    assert tuple(res.fmap.shape[1:]) == (256, 21, 21)
    # D_nodes' first layer on RoI features with 1 704 one-hot planes against torch's convolution of the same (spectrally normalised) weight
    x = gan._roi_planes(res.node_feat.float(), dev_batch[4][:, 1], NOBJ)
    gan.eval()
    with torch.no_grad():
        got = gan.D_nodes[0](x)
        w = gan.D_nodes[0].effective_weight()
        # (float64 on the host: the reference of this check must not be another GPU library's algorithm choice)
        want = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), gan.D_nodes[0].bias.double().cpu()).permute(0, 2, 3, 1)
        scale = float(want.abs().max())
        assert float((got.double().cpu() - want).abs().max()) <= 2e-5 * scale, (float((got.double().cpu() - want).abs().max()), scale)
        # ... and the form the losses use (class planes as a table lookup, GAN._score_rois) against the plain stack on the one-hot planes
        full = gan.D_nodes(x)
        fast = gan._score_rois(gan.D_nodes, res.node_feat.float(), dev_batch[4][:, 1], NOBJ)
        assert torch.isfinite(full).all() and float((fast - full).abs().max()) <= 2e-5 * float(full.abs().max()), (fast.flatten()[:4], full.flatten()[:4])
    gan.train()
    sgg_params = [p for p in model.parameters() if p.requires_grad]
    opt = torch.optim.SGD(sgg_params, lr=1e-3, momentum=0.9)
    G_opt = torch.optim.Adam([p for n, p in gan.named_parameters() if n.startswith('G_')], lr=1e-4, betas=(0.5, 0.999))
    D_opt = torch.optim.Adam([p for n, p in gan.named_parameters() if n.startswith('D_')], lr=1e-4, betas=(0.5, 0.999))
    watch = {'G_obj_embed.weight': gan.G_obj_embed.weight, 'D_nodes.0': gan.D_nodes[0].weight_orig if hasattr(gan.D_nodes[0], 'weight_orig') else gan.D_nodes[0].weight,
             'rel_fc.weight': model.rel_fc.weight, 'obj_fc.weight': model.obj_fc.weight}
    before = {k: v.detach().clone() for k, v in watch.items()}
    losses = gan_train_step(model, gan, res, dev_batch[3].clone(), dev_batch[4].clone(), dev_batch[5].clone(), opt, G_opt, D_opt)
    torch.cuda.synchronize()
    assert sorted(losses) == ['D_fmap', 'D_obj', 'D_rel', 'G_fmap', 'G_obj', 'G_rel', 'rec'], sorted(losses)
    assert all(np.isfinite(float(v)) for v in losses.values()), losses
    for k, v in watch.items():
        assert float((v.detach() - before[k]).abs().max()) > 0, k
