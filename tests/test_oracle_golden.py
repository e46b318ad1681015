"""CPU: the oracle (oracle/sgg_oracle.py) against golden vectors produced by the reference's own code
(tests/golden/make_golden.py).  This is what PINS the oracle."""
import numpy as np
import pytest
import torch

from oracle import sgg_oracle as O
from tests.conftest import weights


def test_raster_bit_exact(golden):
    g = golden('raster')
    np.testing.assert_array_equal(O.draw_union_boxes(g['pairs'], 27), g['out27'])
    np.testing.assert_array_equal(O.draw_union_boxes(g['pairs'][:16], 7), g['out7'])


def test_raster_degenerate_is_nan_like_reference():
    # zero-width union: division by zero -> inf/NaN in the reference too (SURVEY a-5)
    out = O.draw_union_boxes(np.array([[5, 5, 5, 9, 5, 6, 5, 8]], np.float32), 27)
    assert not np.isfinite(out).all()


def test_raster_rejects_bad_shape():
    with pytest.raises(ValueError):
        O.draw_union_boxes(np.zeros((3, 4), np.float32), 27)


@pytest.mark.parametrize('tag', ['d128', 'd32'])
def test_union_boxes_and_feats(golden, tag):
    g = golden('union_feats')
    out = O.union_boxes_and_feats(torch.from_numpy(g[tag + '_pools']), g[tag + '_rois'], g[tag + '_union_inds'],
                                  weights(g, tag))
    np.testing.assert_allclose(out.numpy(), g[tag + '_out'], atol=2e-6, rtol=1e-6)


def test_gru_cell(golden):
    g = golden('gru')
    w = weights(g, '')  # keys 'w_weight_ih' -> prefix '_w_' does not apply; build by hand
    w = {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith('w_')}
    x, h = torch.from_numpy(g['x']), torch.from_numpy(g['h'])
    out = O.gru_cell(x, h, w['weight_ih'], w['weight_hh'], w['bias_ih'], w['bias_hh'])
    np.testing.assert_allclose(out.numpy(), g['out'], atol=1e-6)
    out0 = O.gru_cell(x, torch.zeros_like(h), w['weight_ih'], w['weight_hh'], w['bias_ih'], w['bias_hh'])
    np.testing.assert_allclose(out0.numpy(), g['out_h0'], atol=1e-6)


@pytest.mark.parametrize('tag', ['h32_b1', 'h64_b3', 'h32_sampled', 'h128_b2'])
def test_message_pass(golden, tag):
    g = golden('message_pass')
    p = weights(g, tag)
    for it in range(4):
        v, e = O.message_pass(g[tag + '_rel_rep'], g[tag + '_obj_rep'], g[tag + '_rel_inds'][:, 1:3], p, mp_iter=it)
        np.testing.assert_allclose(v.numpy(), g['%s_v%d' % (tag, it)], atol=2e-6)
        np.testing.assert_allclose(e.numpy(), g['%s_e%d' % (tag, it)], atol=2e-6)


@pytest.mark.parametrize('tag', ['small', 'b3'])
def test_predict(golden, tag):
    g = golden('predict')
    od, rd = O.predict(g[tag + '_node_feat'], g[tag + '_edge_feat'], g[tag + '_rel_inds'], g[tag + '_rois'],
                       weights(g, tag))
    np.testing.assert_allclose(od.numpy(), g[tag + '_obj_dists'], atol=5e-6)
    np.testing.assert_allclose(rd.numpy(), g[tag + '_rel_dists'], atol=5e-6)


@pytest.mark.parametrize('tag', ['b1', 'b3', 'b8x32'])
def test_pair_indexing(golden, tag):
    g = golden('pairs')
    for ov in (0, 1):
        ri = O.get_rel_inds_eval(g[tag + '_im_inds'], g[tag + '_boxes'], bool(ov))
        np.testing.assert_array_equal(ri, g['%s_eval_ov%d' % (tag, ov)])
    im = g[tag + '_im_inds']
    rois = np.concatenate((im[:, None].astype(np.float32), g[tag + '_boxes']), 1)
    _, lab, rl = O.proposal_assignments_gtbox(rois, g[tag + '_boxes'], g[tag + '_gt_classes'], g[tag + '_gt_rels'])
    np.testing.assert_array_equal(lab, g[tag + '_train_labels'])
    np.testing.assert_array_equal(rl, g[tag + '_train_rel_labels'])
    np.testing.assert_array_equal(rl[:, :3], g[tag + '_train_rel_inds'])


def test_eval_tail(golden):
    g = golden('eval_tail')
    for mode, pre in (('sgcls', 'sg_'), ('predcls', 'pc_')):
        b, c, s, r, ps = O.eval_tail(g['obj_dists'], g['rel_dists'], g['rel_inds'], g['boxes'], mode, g['gt_classes'])
        np.testing.assert_array_equal(b, g[pre + 'boxes'])
        np.testing.assert_array_equal(c, g[pre + 'classes'])
        np.testing.assert_allclose(s, g[pre + 'scores'], atol=1e-7)
        np.testing.assert_array_equal(r, g[pre + 'rels'])
        np.testing.assert_allclose(ps, g[pre + 'pred_scores'], atol=1e-7)


def test_losses(golden):
    g = golden('losses')
    np.testing.assert_allclose(O.node_losses(g['obj_dists'], g['obj_labels']).numpy(), g['node'], rtol=1e-6)
    for lt in ('baseline', 'dnorm', 'dnorm-fgbg'):
        np.testing.assert_allclose(O.edge_losses(g['rel_dists'], g['rel_labels'], lt).numpy(), g['edge_' + lt], rtol=1e-6)
        np.testing.assert_allclose(O.edge_losses(g['rel_dists'], np.zeros_like(g['rel_labels']), lt).numpy(),
                                   g['edge_nofg_' + lt], rtol=1e-6)


# ---------------- [3P] rows with no reference-side pin: analytic self-consistency ----------------
def test_roi_align_constant_and_ramp():
    H = W = 38
    const = np.full((2, 3, H, W), 2.5, np.float32)
    rois = np.array([[0, 16, 32, 300, 200], [1, 0, 0, 591, 591], [1, 100.3, 50.7, 130.2, 400.9]], np.float32)
    out = O.roi_align(const, rois)
    np.testing.assert_allclose(out, 2.5, rtol=1e-6)
    # linear ramp f(y,x) = 3x + 2y: bilinear interpolation is exact inside the map, so each bin equals the ramp at
    # the bin centre (mean of symmetric samples)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing='ij')
    ramp = np.broadcast_to((3 * xx + 2 * yy)[None, None], (1, 1, H, W)).astype(np.float32)
    roi = np.array([[0, 64, 96, 400, 320]], np.float32)
    out = O.roi_align(ramp, roi)[0, 0]
    x1, y1, x2, y2 = roi[0, 1:] / 16
    bw, bh = (x2 - x1) / 7, (y2 - y1) / 7
    cx = x1 + (np.arange(7) + .5) * bw
    cy = y1 + (np.arange(7) + .5) * bh
    np.testing.assert_allclose(out, 3 * cx[None] + 2 * cy[:, None], rtol=1e-5)


def test_roi_align_min_size_and_outside():
    fm = np.random.RandomState(0).rand(1, 2, 38, 38).astype(np.float32)
    # degenerate roi -> width/height forced to 1 (aligned=False)
    out = O.roi_align(fm, np.array([[0, 80, 80, 80, 80]], np.float32))
    assert np.isfinite(out).all() and out.std() > 0
    # roi entirely outside -> zeros
    out = O.roi_align(fm, np.array([[0, 700, 700, 900, 900]], np.float32))
    np.testing.assert_array_equal(out, 0)


def test_transform_identity_at_592_and_pad():
    im = torch.rand(3, 592, 592)
    batch, sizes, boxes = O.transform([im], [np.array([[1., 2., 3., 4.]], np.float32)])
    assert batch.shape == (1, 3, 608, 608) and sizes == [(592, 592)]
    mean = torch.tensor(O.IMAGENET_MEAN).view(3, 1, 1)
    std = torch.tensor(O.IMAGENET_STD).view(3, 1, 1)
    torch.testing.assert_close(batch[0, :, :592, :592], (im - mean) / std)
    assert float(batch[0, :, 592:].abs().max()) == 0 and float(batch[0, :, :, 592:].abs().max()) == 0
    np.testing.assert_array_equal(boxes[0].numpy(), [[1, 2, 3, 4]])
    # non-592 input gets resized so that the long side is 592
    b2, s2, bx2 = O.transform([torch.rand(3, 296, 296)], [np.array([[10., 10., 20., 20.]], np.float32)])
    assert s2 == [(592, 592)] and b2.shape[-1] == 608
    np.testing.assert_allclose(bx2[0].numpy(), [[20, 20, 40, 40]])


def test_vgg_names():
    assert O.vgg16_layer_names() == [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]


def _recall_cases(d):
    n = int(d['n_cases'])
    return [{k[len('c%d_' % i):]: d[k] for k in d if k.startswith('c%d_' % i)} for i in range(n)]


@pytest.mark.parametrize('mode', ['predcls', 'sgcls', 'sgdet', 'phrdet', 'objcls'])
@pytest.mark.parametrize('mp', [0, 1])
def test_recall_oracle_matches_reference_evaluator(golden, mode, mp):
    """lib/sgg_eval.py run in the build container (tests/golden/make_golden.py: gold_recall) vs the oracle's restatement."""
    d = golden('recall')
    firsts, nmatch, recs = [], [], {k: [] for k in (20, 50, 100, 200, 300)}
    for c in _recall_cases(d):
        rec, first, m = O.recall_entry(c, c, mode, bool(mp))
        firsts.append(first)
        nmatch.append(m)
        for k in recs:
            recs[k].append(rec[k])
    for k in recs:
        np.testing.assert_allclose(recs[k], d['recall_%s_%d_%d' % (mode, mp, k)], rtol=0, atol=1e-12)
    np.testing.assert_array_equal(nmatch, d['nmatch_%s_%d' % (mode, mp)])
    got, exp = np.concatenate(firsts), d['first_%s_%d' % (mode, mp)]
    if (mode, mp) == ('objcls', 1):     # one-hot predicate scores: pairs (a,b) / (b,a) tie exactly, numpy's argsort order is unspecified
        np.testing.assert_array_equal(got == 2 ** 31 - 1, exp == 2 ** 31 - 1)
    else:
        np.testing.assert_array_equal(got, exp)


def test_freq_bias_oracle_equals_reference(golden):
    """lib/get_dataset_counts.py + lib/sparse_targets.py + rel_model_stanford.py:159-177 (tests/golden/freq_bias.npz)"""
    g = golden('freq_bias')
    n, C, P = int(g['n_img']), int(g['n_cls']), int(g['n_pred'])
    cls = [g['classes_%d' % i] for i in range(n)]
    rels = [g['rels_%d' % i] for i in range(n)]
    boxes = [g['boxes_%d' % i] for i in range(n)]
    for ov in (True, False):
        fg, bg = O.get_counts(cls, rels, boxes, C, P, ov)
        np.testing.assert_array_equal(fg, g['fg_%d' % ov])
        np.testing.assert_array_equal(bg, g['bg_%d' % ov])
    fg, bg = O.get_counts(cls, rels, boxes, C, P, True)
    np.testing.assert_allclose(O.freq_bias_table(fg, bg), g['table'], rtol=0, atol=1e-7)
    for mode in ('sgcls', 'predcls'):
        for tb in (False, True):
            preds, out = O.freq_bias_apply(g['obj_dists'], g['rel_dists'], g['rel_inds'], g['table'], C, mode, g['gt_classes'], tb)
            np.testing.assert_array_equal(preds, g['preds_%s' % mode])
            np.testing.assert_allclose(out, g['out_%s_%d' % (mode, tb)], rtol=0, atol=1e-7)


def test_raw_boxes_raster_and_feats(golden):
    """edge_model 'raw_boxes': lib/get_union_boxes.py:69-116 (grid_sample raster) and the module's eval forward"""
    g = golden('union_feats')
    ims = [tuple(x) for x in g['raw_im_sizes']]
    r = O.draw_union_boxes_grid(g['raw_rois'], g['raw_union_inds'], ims, 27)
    np.testing.assert_allclose(r, g['raw_rects'], atol=1e-5)
    assert r.min() == 0 and r.max() == 1 and ((r > 0) & (r < 1)).any()          # soft edges exist
    p = {k[len('raw_w_'):]: torch.from_numpy(g[k]) for k in g.keys() if k.startswith('raw_w_')}
    out = O.union_boxes_and_feats(g['raw_pools'], g['raw_rois'], g['raw_union_inds'], p, edge_model='raw_boxes', im_sizes=ims)
    np.testing.assert_allclose(out.numpy(), g['raw_out'], atol=2e-5)


def test_rel_assignments_oracle_equals_reference():
    """lib/rel_assignments.py:12 (sgdet training): same numpy seed -> the same sampled rows, bit for bit, incl. the FG cap
    (b2dense), images where nothing matches (b2none) and both filter_non_overlap settings."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'rel_assign.npz'))
    for tag in g['cases']:
        for k in range(5):
            per_gt, nonov, seed = [int(v) for v in g['%s_cfg%d' % (tag, k)]]
            np.random.seed(seed)
            out = O.rel_assignments(g[tag + '_im_inds'], g[tag + '_boxes'], g[tag + '_labels'], g[tag + '_gt_boxes'],
                                    g[tag + '_gt_classes'], g[tag + '_gt_rels'], 0, num_sample_per_gt=per_gt,
                                    filter_non_overlap=bool(nonov))
            np.testing.assert_array_equal(out, g['%s_out%d' % (tag, k)], err_msg='%s case %d' % (tag, k))


def test_gan_ops_oracle_equals_reference():
    """augment/layout.py:33 boxes_to_layout and augment/graphconv.py:17 GraphTripleConv vs the reference's outputs."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'gan_ops.npz'))
    for tag in ('patch', 'vec'):
        for hw in ((38, 38), (10, 14)):
            for pool in ('sum', 'avg'):
                key = 'lay_%s_%dx%d_%s' % (tag, hw[0], hw[1], pool)
                out = O.boxes_to_layout(g['lay_%s_in' % tag], g['lay_boxes'], g['lay_img'], hw[0], hw[1], pool)
                np.testing.assert_allclose(out, g[key + '_out'], atol=2e-5, err_msg=key)
                assert np.abs(out[1]).max() == 0                 # the image without objects
    for k in range(3):
        final, avg, dout = [int(v) for v in g['gc%d_cfg' % k]]
        p = {n[len('gc%d_' % k):]: g[n] for n in g.files if n.startswith('gc%d_net' % k)}
        no, npred = O.graph_triple_conv(g['gc_obj'], g['gc_pred'], g['gc_edges'], p, int(g['gc_hidden']), 'avg' if avg else 'sum', bool(final))
        np.testing.assert_allclose(no, g['gc%d_out_obj' % k], atol=2e-5)
        np.testing.assert_allclose(npred, g['gc%d_out_pred' % k], atol=2e-5)
