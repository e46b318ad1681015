"""GPU parity of the whole drop-in path: RelModelStanford.forward() on the HIP library vs the CPU oracle
chained in the reference's order (oracle.forward_gtbox), on identical seeded inputs and weights.

Tolerances: fp32 mode 1e-3 absolute on obj_dists / rel_dists (north_star).  bf16 mode is checked on the
quantities R@K depends on (ranking of triples), not on raw logits."""
import numpy as np
import pytest
import torch

from oracle import sgg_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def setup():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import sgg_amd
    from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
    S = 160
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls', min_size=S, max_size=S))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(DEV).eval()
    batch = synthetic_batch(B=3, S=S, n_boxes=7, n_fg=3, seed=5, ragged=True)
    with torch.no_grad():
        ref = O.forward_gtbox(batch[0], batch[3], batch[4], batch[5], sd, mode='sgcls', min_size=S, max_size=S)
    return model, batch, ref, sd


def run(model, batch, dtype, mode='sgcls', train=False):
    model.set_compute_dtype(dtype)
    model.mode = mode
    model.train(train)
    with torch.no_grad():
        out = model([batch])
    model.eval()
    return out


def test_forward_fp32_eval_matches_oracle(setup):
    model, batch, ref, _ = setup
    boxes, cls, scores, rels, pred_scores = run(model, batch, torch.float32)
    rb, rc, rs, rr, rp = ref['dets']
    np.testing.assert_array_equal(boxes, rb)
    np.testing.assert_array_equal(cls, rc)
    np.testing.assert_allclose(scores, rs, atol=1e-3)
    np.testing.assert_array_equal(rels, rr)
    np.testing.assert_allclose(pred_scores, rp, atol=1e-3)


def test_forward_fp32_intermediates_within_1e3(setup):
    model, batch, ref, _ = setup
    model.set_compute_dtype(torch.float32)
    model.mode = 'sgcls'
    with torch.no_grad():
        gt_boxes, gt_classes = batch[3].to(DEV), batch[4].to(DEV)
        res = model.faster_rcnn(batch[0], gt_boxes, gt_classes, None)
        np.testing.assert_allclose(res.fmap.float().cpu().numpy(), ref['fmap'].numpy(), atol=1e-3, rtol=1e-3)
        rel_inds = model.get_rel_inds(None, res.im_inds, res.rm_box_priors)
        np.testing.assert_array_equal(rel_inds.cpu().numpy(), ref['rel_inds'])
        rois = torch.cat((res.im_inds[:, None].float(), res.rm_box_priors), 1)
        nf, ef = model.node_edge_features(res.fmap, rois, rel_inds[:, 1:], res.im_sizes)
        assert tuple(nf.shape) == ref['node_feat'].shape and tuple(ef.shape) == ref['edge_feat'].shape
        np.testing.assert_allclose(nf.float().cpu().numpy(), ref['node_feat'], atol=1e-3, rtol=1e-3)
        np.testing.assert_allclose(ef.float().cpu().numpy(), ref['edge_feat'], atol=1e-3, rtol=1e-3)
        od, rd = model.predict(nf, ef, rel_inds, rois, res.im_sizes)
        np.testing.assert_allclose(od.cpu().numpy(), ref['rm_obj_dists'].numpy(), atol=1e-3)
        np.testing.assert_allclose(rd.cpu().numpy(), ref['rel_dists'].numpy(), atol=1e-3)
        # reference-layout (plain NCHW) inputs from another producer give the same answer
        od2, rd2 = model.predict(nf.contiguous(), ef.contiguous(), rel_inds, rois, res.im_sizes)
        np.testing.assert_allclose(od2.cpu().numpy(), od.cpu().numpy(), atol=1e-5)
        np.testing.assert_allclose(rd2.cpu().numpy(), rd.cpu().numpy(), atol=1e-5)
        # UnionBoxesAndFeats.forward == union_pools + conv(rects)
        ub = model.union_boxes(ef, rois, rel_inds[:, 1:], res.im_sizes)
        up = {k[len('union_boxes.'):]: v for k, v in setup[3].items() if k.startswith('union_boxes.')}
        exp = O.union_boxes_and_feats(torch.from_numpy(ref['edge_feat']), rois.cpu().numpy(),
                                      rel_inds[:, 1:].cpu().numpy(), up)
        np.testing.assert_allclose(ub.float().cpu().numpy(), exp.numpy(), atol=1e-3, rtol=1e-3)


def test_forward_predcls_and_result_fields(setup):
    model, batch, ref, sd = setup
    out = run(model, batch, torch.float32, mode='predcls')
    with torch.no_grad():
        refp = O.eval_tail(ref['rm_obj_dists'], ref['rel_dists'], ref['rel_inds'], batch[3].numpy(), 'predcls',
                           batch[4][:, 1].numpy())
    np.testing.assert_array_equal(out[1], refp[1])
    np.testing.assert_array_equal(out[2], np.ones(len(refp[1]), np.float32))
    np.testing.assert_array_equal(out[3], refp[3])
    np.testing.assert_allclose(out[4], refp[4], atol=1e-3)


def test_forward_bf16_ranking_close_to_oracle(setup):
    model, batch, ref, _ = setup
    boxes, cls, scores, rels, pred_scores = run(model, batch, torch.bfloat16)
    rb, rc, rs, rr, rp = ref['dets']
    assert (cls == rc).mean() >= 0.85                 # argmax object class mostly unchanged by bf16
    np.testing.assert_allclose(scores, rs, atol=0.08)
    probs = torch.softmax(torch.as_tensor(ref['rm_obj_dists']).float(), 1)
    probs[:, 0] = 0
    chosen = probs[torch.arange(len(cls)), torch.as_tensor(cls)].numpy()
    assert (chosen >= 0.5 * rs).all()                 # a differing argmax is a close runner-up, never a far-off class
    # top-K triple sets overlap (what R@K consumes)
    K = 50
    top = set(map(tuple, rels[:K]))
    rtop = set(map(tuple, rr[:K]))
    assert len(top & rtop) >= int(0.7 * K)


def test_state_dict_round_trip_and_reference_keys(setup):
    model, _, _, sd = setup
    for k in ('detector.backbone.0.weight', 'detector.backbone.28.bias', 'roi_fmap.1.0.weight', 'roi_fmap.1.3.bias',
              'roi_fmap_obj.0.weight', 'roi_fmap_obj.3.bias', 'union_boxes.conv.0.weight',
              'union_boxes.conv.6.running_var', 'obj_unary.weight', 'edge_unary.bias', 'edge_gru.weight_ih',
              'node_gru.bias_hh', 'sub_vert_w_fc.0.weight', 'in_edge_w_fc.0.bias', 'obj_fc.weight', 'rel_fc.bias',
              'detector.roi_heads.box_head.fc6.weight', 'detector.rpn.head.conv.weight'):
        assert k in sd, k
    assert sd['roi_fmap.1.0.weight'].shape == (4096, 25088) and sd['edge_gru.weight_ih'].shape == (1536, 512)
    model.load_state_dict(sd)


def test_error_contract(setup):
    model, batch, _, _ = setup
    with pytest.raises(AssertionError):
        model([batch, batch])                       # rel_model_stanford.py:121
    model.mode = 'bogus'
    with pytest.raises(NotImplementedError):
        with torch.no_grad():
            model([batch])                          # rel_model_stanford.py:193
    model.mode = 'sgcls'


def test_train_relation_subsampling_keeps_order_and_caps(setup):
    """lib/proposal_assignments_gtbox.py:47-66: more candidates than RELS_PER_IMG -> random subset, still sorted."""
    model, batch, _, _ = setup
    model.train()
    old = model.RELS_PER_IMG
    model.RELS_PER_IMG = 8                      # 3 images: <= 6 FG rows, 24 rows in total
    try:
        gt_boxes, gt_classes, gt_rels = batch[3].to(DEV), batch[4].to(DEV), batch[5].to(DEV)
        torch.manual_seed(0)
        _, labels, rl = model.gt_labels(gt_boxes, gt_classes, gt_rels)
        rl = rl.cpu().numpy()
        assert len(rl) == 24 and (rl[:, 3] > 0).sum() == 6
        key = rl[:, 0] * 10 ** 6 + rl[:, 1] * 1000 + rl[:, 2]
        assert (np.diff(key) >= 0).all()                                  # still sorted by (img, subj, obj)
        full = O.proposal_assignments_gtbox(np.concatenate((batch[4][:, :1].float().numpy(), batch[3].numpy()), 1),
                                            batch[3].numpy(), batch[4].numpy(), batch[5].numpy(), RELS_PER_IMG=10 ** 6)[2]
        rows = set(map(tuple, full.tolist()))
        assert all(tuple(r) in rows for r in rl.tolist())                 # a subset of the un-sampled assignment
        _, _, rl2 = model.gt_labels(gt_boxes, gt_classes, gt_rels)
        assert not np.array_equal(rl, rl2.cpu().numpy())                 # random subset
    finally:
        model.RELS_PER_IMG = old
        model.eval()


def test_full_size_forward_fp32_matches_oracle_and_bf16_batch_is_sane():
    """BASELINE-size inputs (592x592 frames, 32 boxes, 992 edges per image): one image in exact-fp32 mode against the
    oracle within the 1e-3 bar, then the benchmark batch (8 images, bf16) through size-independent properties -- finite,
    triple scores sorted, every image's 992 ordered pairs present exactly once."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import sgg_amd
    from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls'))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(DEV).eval()
    b1 = synthetic_batch(B=1, S=592, n_boxes=32, n_fg=6, seed=77)
    torch.set_num_threads(8)
    with torch.no_grad():
        ref = O.forward_gtbox(b1[0], b1[3], b1[4], b1[5], sd, mode='sgcls')
    model.set_compute_dtype(torch.float32)
    model.train()                                   # Result with the raw distributions (eval would only give the tuple)
    model.dropout_p = 0.0
    with torch.no_grad():
        model.eval()
        boxes, cls, scores, rels, pred_scores = model([b1])
    rb, rc, rs, rr, rp = ref['dets']
    np.testing.assert_allclose(scores, rs, atol=1e-3)
    assert (cls == rc).all()
    # same triples; rows are rank-ordered, so compare as {pair: scores} maps
    got = {tuple(r): p for r, p in zip(rels.tolist(), pred_scores)}
    exp = {tuple(r): p for r, p in zip(rr.tolist(), rp)}
    assert got.keys() == exp.keys() and len(got) == 992
    worst = max(float(np.abs(got[k] - exp[k]).max()) for k in exp)
    assert worst <= 1e-3, worst
    # the benchmark batch in bf16
    model.set_compute_dtype(torch.bfloat16)
    model.eval()
    b8 = synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=111)
    with torch.no_grad():
        boxes, cls, scores, rels, pred_scores = model([b8])
    assert boxes.shape == (256, 4) and rels.shape == (7936, 2) and pred_scores.shape == (7936, 51)
    assert np.isfinite(pred_scores).all() and np.isfinite(scores).all()
    np.testing.assert_allclose(pred_scores.sum(1), 1.0, atol=2e-2)
    trip = pred_scores[:, 1:].max(1) * scores[rels[:, 0]] * scores[rels[:, 1]]
    assert (np.diff(trip) <= 1e-6).all()                                       # rank order of filter_dets
    pairs = set(map(tuple, rels.tolist()))
    assert len(pairs) == 7936 and all(s // 32 == o // 32 and s != o for s, o in pairs)
    model.dropout_p = 0.5
    del model
    torch.cuda.empty_cache()


def test_forward_raw_boxes_edge_model_matches_oracle():
    """-edge_model raw_boxes (config.py:179 lists it next to the default 'motifs'): eval tuple in fp32 vs the oracle, and a train
    step through the same raster."""
    import sgg_amd
    from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
    from sgg_amd.trainer import Trainer
    S = 160
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls', min_size=S, max_size=S, edge_model='raw_boxes'))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(DEV).eval()
    model.set_compute_dtype(torch.float32)
    batch = synthetic_batch(B=2, S=S, n_boxes=6, n_fg=3, seed=12, ragged=True)
    with torch.no_grad():
        ref = O.forward_gtbox(batch[0], batch[3], batch[4], batch[5], sd, mode='sgcls', min_size=S, max_size=S, edge_model='raw_boxes')
        plain = O.forward_gtbox(batch[0], batch[3], batch[4], batch[5], sd, mode='sgcls', min_size=S, max_size=S)
        boxes, cls, scores, rels, pred_scores = model([batch])
    assert float((ref['rel_dists'] - plain['rel_dists']).abs().max()) > 1e-3      # the two rasters do differ
    rb, rc, rs, rr, rp = ref['dets']
    np.testing.assert_array_equal(cls, rc)
    np.testing.assert_allclose(scores, rs, atol=1e-3)
    np.testing.assert_array_equal(rels, rr)
    np.testing.assert_allclose(pred_scores, rp, atol=1e-3)
    tr = Trainer(model, lr=1e-3)
    l0 = float(tr.step(batch))
    l1 = float(tr.step(batch))
    assert np.isfinite(l0) and np.isfinite(l1)


def test_get_scaled_boxes_matches_reference_formula(setup):
    """rel_model_base.py:262-274 (the GAN / feature-extraction callers, main.py:137): per-image division by (w, h) and the <= 1 check"""
    model, batch, ref, _ = setup
    boxes = torch.tensor([[10., 20., 80., 90.], [0., 0., 159., 99.], [5., 5., 50., 60.]], device=DEV)
    im_inds = torch.tensor([0, 0, 1], device=DEV)
    im_sizes = [(100, 160), (80, 60)]                    # (h, w)
    got = model.get_scaled_boxes(boxes, im_inds, im_sizes).cpu().numpy()
    want = boxes.cpu().numpy().copy()
    want[:2, [0, 2]] /= 160; want[:2, [1, 3]] /= 100
    want[2, [0, 2]] /= 60; want[2, [1, 3]] /= 80
    np.testing.assert_allclose(got, want, rtol=1e-6)
    with pytest.raises(AssertionError):
        model.get_scaled_boxes(boxes * 3, im_inds, im_sizes)


def test_node_edge_features_differentiable_in_fmap(setup):
    """main.py:141-149 (GAN path): node_edge_features on a feature map that requires grad -- the gradient equals the adjoint applied to
    the upstream gradients, and it flows on through predict()'s autograd node in training mode."""
    model, batch, ref, _ = setup
    from sgg_amd import ops
    model.set_compute_dtype(torch.float32)
    g = torch.Generator().manual_seed(2)
    B, C, Hf = 2, 512, 10
    fmap = torch.randn(B, C, Hf, Hf, generator=g).to(DEV).requires_grad_(True)
    rois = torch.tensor([[0, 8., 8., 100., 90.], [0, 30., 20., 150., 140.], [1, 0., 0., 159., 159.], [1, 40., 60., 90., 120.]], device=DEV)
    ui = torch.tensor([[0, 1], [1, 0], [2, 3], [3, 2]], device=DEV)
    node, edge = model.node_edge_features(fmap, rois, ui, [(160, 160), (160, 160)])
    assert node.requires_grad and edge.requires_grad
    gn, ge = torch.randn(node.shape, generator=g).to(DEV), torch.randn(edge.shape, generator=g).to(DEV)
    (node * gn).sum().backward(retain_graph=True)
    d1 = fmap.grad.clone()
    fmap.grad = None
    ((node * gn).sum() + (edge * ge).sum()).backward()
    shape = (B, Hf, Hf, C)
    want = ops.roi_align_bwd(gn.contiguous(), shape, rois, None, 1.0 / 16)
    torch.testing.assert_close(d1, want.permute(0, 3, 1, 2), rtol=1e-5, atol=1e-5)
    ops.roi_align_bwd(ge.contiguous(), shape, rois, ui, 1.0 / 16, d_fmap=want)
    torch.testing.assert_close(fmap.grad, want.permute(0, 3, 1, 2), rtol=1e-5, atol=1e-5)
    with torch.no_grad():                                       # no graph requested: plain kernels, same numbers
        n2, e2 = model.node_edge_features(fmap, rois, ui, [(160, 160), (160, 160)])
    assert not n2.requires_grad and torch.equal(n2, node.detach()) and torch.equal(e2, edge.detach())
