"""ResNet-50-FPN front end of the GQA configuration (sgg_models/rel_model_base.py:58-81) on the HIP path against the
oracle's restatement of torchvision's backbone ([3P] unpinned), and its glue kernels against plain torch ops."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import sgg_oracle as O  # noqa: E402

DEV = 'cuda:0'


def seeded_detector(seed=0):
    """random He-scaled convolutions, non-trivial frozen BatchNorm buffers"""
    from sgg_amd.resnet_fpn import FrozenBatchNorm2d, ResNet50FPNDetector
    torch.manual_seed(seed)
    det = ResNet50FPNDetector(num_classes=12)
    for m in det.modules():
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.kaiming_normal_(m.weight, nonlinearity='relu')
            if m.bias is not None:
                torch.nn.init.normal_(m.bias, std=0.1)
        elif isinstance(m, FrozenBatchNorm2d):
            m.weight.uniform_(0.4, 0.9)
            m.bias.normal_(0, 0.1)
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
    return det


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_glue_kernels_equal_torch(dtype):
    from sgg_amd import ops
    g = torch.Generator().manual_seed(3)
    tol = dict(atol=0, rtol=0)
    x = torch.randn(2, 11, 14, 16, generator=g).to(DEV).to(dtype)                   # NHWC rows
    nchw = lambda t: t.permute(0, 3, 1, 2).float()                                   # noqa: E731
    torch.testing.assert_close(nchw(ops.maxpool3x3s2(x)), F.max_pool2d(nchw(x), 3, 2, 1), **tol)
    y = torch.randn(2, 11, 14, 16, generator=g).to(DEV).to(dtype)
    torch.testing.assert_close(ops.add_relu_(y.clone(), x).float(), (y.float() + x.float()).clamp(min=0).to(dtype).float(), **tol)
    # rows -> zero-bordered plane, stride-2 sub-sampling
    pl = torch.zeros(2, 13, 16, 16, dtype=dtype, device=DEV)
    ops.plane_copy(x, pl, dst_pad=1)
    assert torch.equal(pl[:, 1:-1, 1:-1], x) and pl[:, 0].abs().max() == 0 and pl[:, :, 0].abs().max() == 0 and pl[:, -1].abs().max() == 0
    sub = ops.plane_copy(x, torch.empty(2, 6, 7, 16, dtype=dtype, device=DEV), stride=2)
    assert torch.equal(sub, x[:, ::2, ::2])
    assert torch.equal(ops.plane_copy(pl, torch.empty(2, 6, 7, 16, dtype=dtype, device=DEV), src_pad=1, stride=2), x[:, ::2, ::2])
    # patch matrices: 3x3 / 2 / 1 on a bordered plane; 7x7 / 2 / 3 on the 4-channel f32 image plane (3 channels used, f32 -> dtype)
    for k, s, p_ in ((3, 2, 1), (3, 1, 1)):
        cols, Ho, Wo = ops.im2col(pl, k, s, p_, src_pad=1)
        ref = F.unfold(nchw(x), k, padding=p_, stride=s)                                   # [B, C*k*k, L], (c, ky, kx) order
        ref = ref.view(2, 16, k * k, Ho * Wo).permute(0, 3, 2, 1).reshape(2 * Ho * Wo, k * k * 16)
        assert cols.shape == (2 * Ho * Wo, 192) and cols[:, 144:].abs().max() == 0
        torch.testing.assert_close(cols[:, :144].float(), ref, **tol)
    img = torch.zeros(2, 21, 26, 4, device=DEV)
    img[:, 1:-1, 1:-1, :3] = torch.randn(2, 19, 24, 3, generator=g).to(DEV)
    cols, Ho, Wo = ops.im2col(img, 7, 2, 3, src_pad=1, C=3, Kp=192, dtype=dtype)
    ref = F.unfold(img[:, 1:-1, 1:-1, :3].permute(0, 3, 1, 2), 7, padding=3, stride=2).view(2, 3, 49, Ho * Wo).permute(0, 3, 2, 1).reshape(-1, 147)
    assert (Ho, Wo) == (10, 12) and cols[:, 147:].abs().max() == 0
    torch.testing.assert_close(cols[:, :147].float(), ref.to(dtype).float(), **tol)


def test_state_dict_names_are_torchvisions():
    from sgg_amd.resnet_fpn import ResNet50FPNDetector
    keys = set(ResNet50FPNDetector(num_classes=1704).state_dict())
    for k in ('backbone.body.conv1.weight', 'backbone.body.bn1.running_var', 'backbone.body.layer1.0.downsample.0.weight',
              'backbone.body.layer1.0.downsample.1.running_mean', 'backbone.body.layer3.5.conv2.weight', 'backbone.body.layer4.2.bn3.bias',
              'backbone.fpn.inner_blocks.0.weight', 'backbone.fpn.layer_blocks.3.bias', 'rpn.head.conv.weight', 'rpn.head.cls_logits.bias',
              'roi_heads.box_head.fc6.weight', 'roi_heads.box_head.fc7.bias', 'roi_heads.box_predictor.cls_score.weight',
              'roi_heads.box_predictor.bbox_pred.bias', 'roi_heads.mask_head.mask_fcn4.weight'):
        assert k in keys, k
    assert not any('num_batches_tracked' in k for k in keys)
    sd = ResNet50FPNDetector(num_classes=1704).state_dict()
    assert sd['roi_heads.box_head.fc6.weight'].shape == (1024, 256 * 49) and sd['roi_heads.box_predictor.cls_score.weight'].shape == (1704, 1024)
    assert sd['rpn.head.cls_logits.weight'].shape[0] == 3 and len([k for k in keys if k.startswith('backbone.body.') and k.endswith('conv2.weight')]) == 16


@pytest.mark.parametrize('dtype,size', [(torch.float32, (96, 128)), (torch.float32, (70, 50)), (torch.bfloat16, (96, 128))])
def test_pool_level_feature_map_matches_oracle(dtype, size):
    """transform (no resize / resize + pad, two images of different sizes) -> ResNet-50 -> FPN top level -> LastLevelMaxPool"""
    det = seeded_detector(1).to(DEV)
    det.transform.min_size, det.transform.max_size = 96, 128
    g = torch.Generator().manual_seed(5)
    images = [torch.rand(3, *size, generator=g), torch.rand(3, size[0] - 12, size[1] - 20, generator=g)]
    fmap, sizes, padded = det.features(images, dtype)
    batch, osizes, _ = O.transform(images, None, 96, 128)
    assert list(map(tuple, osizes)) == list(map(tuple, sizes)) and tuple(batch.shape[2:]) == tuple(padded)
    sd = {'detector.' + k: v.cpu() for k, v in det.state_dict().items()}
    want = O.resnet50_fpn_pool(batch, sd)                                             # [B,256,h,w]
    got = fmap.float().cpu().permute(0, 3, 1, 2)
    assert got.shape == want.shape and want.abs().max() > 0.05
    err = (got - want).abs().max().item() / want.abs().max().item()
    assert err <= (2e-4 if dtype == torch.float32 else 6e-2), err


def test_relation_model_with_resnet50_backbone_matches_oracle():
    """RelModelStanford(backbone='resnet50') (the GQA configuration, rel_model_base.py:58-81): obj_dim 1024, 256-channel 'pool'-level
    feature map at stride 64, TwoMLPHead RoI heads (ReLU after the edge branch's fc7 too); eval forward in exact-fp32 mode against the
    oracle chain (transform -> ResNet-50-FPN -> RoIAlign at 1/64 -> heads -> message passing -> eval tail)."""
    import sgg_amd
    from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
    from sgg_amd.resnet_fpn import FrozenBatchNorm2d
    torch.manual_seed(2)
    S = 256
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls', backbone='resnet50', min_size=S, max_size=S))
    for m in model.modules():
        if isinstance(m, FrozenBatchNorm2d):
            m.weight.uniform_(0.5, 1.0)
            m.running_var.uniform_(0.6, 1.4)
            m.running_mean.normal_(0, 0.1)
    assert model.obj_dim == 1024 and model.edge_dim == 256 and model.fmap_sz == 21
    sd = model.state_dict()
    for k, shape in (('roi_fmap.fc6.weight', (1024, 256 * 49)), ('roi_fmap_obj.fc7.weight', (1024, 1024)), ('obj_unary.weight', (512, 1024)),
                     ('union_boxes.conv.4.weight', (256, 128, 3, 3)), ('detector.backbone.fpn.layer_blocks.3.weight', (256, 256, 3, 3))):
        assert tuple(sd[k].shape) == shape, (k, tuple(sd[k].shape))
    model.to(DEV).eval()
    model.set_compute_dtype(torch.float32)
    batch = synthetic_batch(B=2, S=S, n_boxes=5, n_fg=3, seed=4)
    dev_batch = tuple(t_.to(DEV) if isinstance(t_, torch.Tensor) else t_ for t_ in batch)
    with torch.no_grad():
        dets = model([dev_batch])
    p = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    ref = O.forward_gtbox(batch[0], batch[3].numpy(), batch[4].numpy(), batch[5].numpy(), p, mode='sgcls', min_size=S, max_size=S)
    assert ref['fmap'].shape[1] == 256 and ref['fmap'].shape[-1] == 4            # 256 / 64
    boxes, obj_classes, obj_scores, rels, pred_scores = dets
    rb, rc, rs, rr, rp = ref['dets']
    np.testing.assert_array_equal(obj_classes, rc)
    np.testing.assert_allclose(obj_scores, rs, atol=1e-3)
    np.testing.assert_array_equal(rels[:20], rr[:20])
    np.testing.assert_allclose(pred_scores[:20], rp[:20], atol=1e-3)
    # ---- training forward and the gradients of every head parameter against torch autograd of the oracle (TwoMLPHead layout:
    # no Dropout, ReLU after the edge branch's fc7), then one Trainer step in bf16
    from sgg_amd.train import param_names
    model.train()
    res = model([dev_batch])
    g = torch.Generator().manual_seed(0)
    Wo, Wr = torch.randn(res.rm_obj_dists.shape, generator=g), torch.randn(res.rel_dists.shape, generator=g)
    model.zero_grad()
    ((res.rm_obj_dists * Wo.to(DEV)).sum() + (res.rel_dists * Wr.to(DEV)).sum()).backward()
    names = param_names(model)
    assert names[:4] == ['roi_fmap.fc6.weight', 'roi_fmap.fc6.bias', 'roi_fmap.fc7.weight', 'roi_fmap.fc7.bias'] and len(names) == 40
    pq = {k: v.clone().requires_grad_(k in set(names)) for k, v in p.items()}
    od, rd = O.predict(res.node_feat.float().cpu().contiguous(), res.edge_feat.float().cpu().contiguous(), res.rel_inds.cpu().numpy(),
                       res.rois.cpu().numpy(), pq, training=True)
    torch.testing.assert_close(res.rm_obj_dists.detach().cpu(), od.detach(), atol=1e-3, rtol=1e-3)
    torch.testing.assert_close(res.rel_dists.detach().cpu(), rd.detach(), atol=1e-3, rtol=1e-3)
    ((od * Wo).sum() + (rd * Wr).sum()).backward()
    named = dict(model.named_parameters())
    for n in names:
        ref, got = pq[n].grad, named[n].grad.cpu()
        err = float((got - ref).abs().max()) / (float(ref.abs().max()) + 1e-6)
        assert got.shape == ref.shape and err < 2e-3, (n, err)
    from sgg_amd.trainer import Trainer
    model.set_compute_dtype(torch.bfloat16)
    tr = Trainer(model, lr=1e-3)
    losses = [float(tr.step(dev_batch)) for _ in range(4)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_pyramid_levels_match_oracle(dtype):
    """P2 .. P5 (lateral 1x1, top-down nearest joins, 3x3 outputs) + the 'pool' level against the oracle's FeaturePyramidNetwork"""
    det = seeded_detector(1).to(DEV)
    g = torch.Generator().manual_seed(5)
    imgs = [torch.rand(3, 150, 200, generator=g), torch.rand(3, 170, 160, generator=g)]
    det.transform.min_size, det.transform.max_size = 192, 256
    with torch.no_grad():
        pool, sizes, padded, levels = det.features([im.to(DEV) for im in imgs], dtype, pyramid=True)
        batch, osizes, _ = O.transform(imgs, None, 192, 256)
        p = {'detector.' + k: v.float().cpu() for k, v in det.state_dict().items()}
        want = O.resnet50_fpn_levels(batch, p)
    assert [tuple(s_) for s_ in sizes] == [tuple(s_) for s_ in osizes] and padded == tuple(batch.shape[-2:])
    assert [lv.shape[1] for lv in levels] == [padded[0] // s_ for s_ in (4, 8, 16, 32)]
    for got, ref in zip(levels + [pool], want):
        got = got.float().permute(0, 3, 1, 2).cpu()
        assert got.shape == ref.shape
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        assert err <= (3e-4 if dtype == torch.float32 else 2e-2), err


def test_upsample_add_equals_interpolate():
    from sgg_amd import ops
    g = torch.Generator().manual_seed(1)
    for (h, w, ht, wt) in ((8, 10, 4, 5), (7, 9, 4, 5), (38, 38, 16, 16)):
        y = torch.randn(2, h, w, 16, generator=g).to(DEV)
        top = torch.randn(2, ht, wt, 16, generator=g).to(DEV)
        want = y + F.interpolate(top.permute(0, 3, 1, 2), size=(h, w), mode='nearest').permute(0, 2, 3, 1)
        assert torch.equal(ops.upsample_add_(y.clone(), top), want)


@pytest.fixture(scope='module')
def sgdet_r50():
    import sgg_amd
    from sgg_amd.resnet_fpn import FrozenBatchNorm2d
    from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
    torch.manual_seed(6)
    S = 192
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgdet', backbone='resnet50', min_size=S, max_size=S))
    for m in model.modules():
        if isinstance(m, FrozenBatchNorm2d):
            m.weight.uniform_(0.5, 1.0)
            m.running_var.uniform_(0.6, 1.4)
            m.running_mean.normal_(0, 0.1)
    with torch.no_grad():        # small regression outputs: proposals stay near their anchors, so NMS has many distinct boxes to order
        model.detector.rpn.head.bbox_pred.weight.mul_(0.01)
        model.detector.roi_heads.box_predictor.bbox_pred.weight.mul_(0.05)
    sd = {k: v.detach().float().clone() for k, v in model.state_dict().items()}
    model.to(DEV).eval().set_compute_dtype(torch.float32)
    model.set_box_score_thresh(0.0)
    batch = synthetic_batch(B=2, S=S, n_boxes=4, n_fg=2, seed=2)
    with torch.no_grad():
        ref = O.sgdet_detect(batch[0], sd, score_thresh=0.0, min_size=S, max_size=S, backbone='resnet50')
    return model, sd, batch, ref, S


def test_fpn_proposals_and_level_assignment_match_oracle(sgdet_r50):
    """RPN over the five maps (3 anchors per location, one size per level), per-level top-1000, NMS inside a level only; LevelMapper;
    four-level RoIAlign -- each against the oracle on the SAME inputs (the HIP path's own pyramid)."""
    from sgg_amd import sgdet
    model, sd, batch, ref, S = sgdet_r50
    with torch.no_grad():
        pool, sizes, padded, levels = model.detector.features([im.to(DEV) for im in batch[0]], torch.float32, pyramid=True)
        w = sgdet.prepared(model, fpn=True)
        img_hw = torch.tensor([[float(s_[0]), float(s_[1])] for s_ in sizes], device=DEV)
        rois, offs = sgdet.propose(model, w, levels + [pool], img_hw, padded)
        nchw = [lv.float().permute(0, 3, 1, 2).cpu() for lv in levels + [pool]]
        want = O.rpn_proposals_fpn(nchw, sd, sizes, padded)
    for b, exp in enumerate(want):
        got = rois[offs[b]:offs[b + 1]].cpu()
        assert (got[:, 0] == b).all() and abs(len(got) - len(exp)) <= max(2, len(exp) // 50)
        # random-init objectness is nearly tied, so compare as sets: almost every proposal has a twin within 0.05 px
        d = (got[:, None, 1:] - exp[None]).abs().amax(2).min(1)[0]
        assert (d < 5e-2).float().mean() >= 0.9, (b, float((d < 5e-2).float().mean()))
    # level assignment and the pooled features of the oracle's own proposals
    props = want[0][:200]
    np.testing.assert_array_equal(sgdet.pyramid_level_of(torch.cat((torch.zeros(len(props), 1), props), 1).to(DEV), 4).cpu().numpy(),
                                  O.fpn_level_of(props).numpy())
    boxes = torch.tensor([[2., 3., 30., 40.], [0., 0., 190., 150.], [50., 20., 170., 140.], [10., 10., 12., 13.]])    # levels 0 .. 2
    top = float(max(s_[0] for s_ in sizes))
    scales = [2.0 ** round(np.log2(float(m.shape[1]) / top)) for m in levels]
    with torch.no_grad():
        feat = sgdet.box_features(model, levels, scales, torch.cat((torch.zeros(4, 1), boxes), 1).to(DEV))
    exp = O.multiscale_roi_align(nchw[:4], boxes, 0, sizes).reshape(4, -1)
    torch.testing.assert_close(feat.float().cpu(), exp, atol=2e-4, rtol=1e-3)


def test_sgdet_with_resnet50_backbone_matches_oracle(sgdet_r50):
    """rel_model_base.py:209-235 with backbone='resnet50' (GQA SGGen): detections as sets (random-init scores are nearly tied, see
    test_sgdet_gpu.py), then the whole forward against the oracle fed with the HIP path's own detections."""
    model, sd, batch, ref, S = sgdet_r50
    with torch.no_grad():
        res = model.faster_rcnn(batch[0], None, batch[4].to(DEV), None)
    im = res.im_inds.cpu().numpy()
    dets = ref[-1]
    for b, (eb, es, el) in enumerate(dets):
        gb = res.rm_box_priors.cpu().numpy()[im == b]
        gl = res.rm_obj_labels.cpu().numpy()[im == b]
        assert len(gb) == len(eb) <= 50
        hit = sum(bool(((np.abs(eb.numpy() - bx[None]).max(1) < 5e-2) & (el.numpy() == lb)).any()) for bx, lb in zip(gb, gl))
        assert hit >= 0.8 * len(gb), (b, hit, len(gb))
    assert tuple(res.fmap.shape[1:]) == (256, 3, 3)
    with torch.no_grad():
        boxes, cls, scores, rels, pred_scores = model([batch])
        exp = O.forward_from_detections(res.fmap.float().cpu(), res.im_inds.cpu().numpy(), res.rm_box_priors.cpu().numpy(),
                                        res.rm_box_priors_org.cpu().numpy(), res.im_sizes, sd)
    rb, rc, rs, rr, rp = exp['dets']
    np.testing.assert_allclose(boxes, rb, atol=1e-5)
    np.testing.assert_array_equal(cls, rc)
    np.testing.assert_allclose(scores, rs, atol=1e-3)
    assert rels.shape == rr.shape
    key = lambda r: r[:, 0] * 100000 + r[:, 1]                                  # noqa: E731
    go, ro = np.argsort(key(rels)), np.argsort(key(rr))
    np.testing.assert_array_equal(rels[go], rr[ro])
    np.testing.assert_allclose(pred_scores[go], rp[ro], atol=1e-3)


def test_sgdet_resnet50_full_size_runs_in_f16():
    """The GQA SGGen configuration at its real size (1333-pixel frames: maps of 336 .. 21 cells, 338 k anchors, <= 4 823 candidates per
    image into the level-wise NMS, 1 000 proposals through the four-level RoIAlign and the 12544 -> 1024 box head) in the default f16
    mode: structure of the output."""
    import sgg_amd
    from sgg_amd.resnet_fpn import FrozenBatchNorm2d
    from sgg_amd.synthetic import SyntheticData, init_weights
    torch.manual_seed(8)
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgdet', backbone='resnet50'))
    for m in model.modules():
        if isinstance(m, FrozenBatchNorm2d):
            m.weight.uniform_(0.5, 1.0)
            m.running_var.uniform_(0.6, 1.4)
    with torch.no_grad():
        model.detector.rpn.head.bbox_pred.weight.mul_(0.01)
        model.detector.roi_heads.box_predictor.bbox_pred.weight.mul_(0.05)
    model.to(DEV).eval()
    assert model.compute_dtype == torch.float16
    model.set_box_score_thresh(0.0)
    g = torch.Generator().manual_seed(1)
    imgs = [torch.rand(3, 1000, 1000, generator=g), torch.rand(3, 900, 1100, generator=g)]
    gt_classes = torch.tensor([[0, 1], [1, 1]])
    batch = (imgs, None, 0, torch.zeros(2, 4), gt_classes, None, None, None)
    with torch.no_grad():
        res = model.faster_rcnn(imgs, None, gt_classes.to(DEV), None)
        assert tuple(res.fmap.shape[1:]) == (256, 21, 21)
        n = [int((res.im_inds == b).sum()) for b in range(2)]
        assert all(2 <= k <= 50 for k in n)
        bx = res.rm_box_priors
        assert (bx[:, 2] > bx[:, 0]).all() and (bx[:, 3] > bx[:, 1]).all() and bx.min() >= 0 and bx[:, 2].max() <= 1333 + 1e-3
        assert (res.rm_obj_labels >= 1).all() and (res.rm_obj_labels < 151).all()
        boxes, cls, scores, rels, pred_scores = model([batch])
    assert boxes.shape[0] == sum(n) == cls.shape[0] and np.isfinite(pred_scores).all() and np.isfinite(scores).all()
    assert rels.shape[1] == 2 and (rels[:, 0] != rels[:, 1]).all() and pred_scores.shape == (rels.shape[0], 51)
    trip = pred_scores[:, 1:].max(1) * scores[rels[:, 0]] * scores[rels[:, 1]]
    assert (trip[:-1] >= trip[1:] - 1e-6).all()
