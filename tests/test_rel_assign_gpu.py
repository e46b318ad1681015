"""SGDet-training relation sampling (lib/rel_assignments.py): HIP tables + host draws against vectors produced by the
reference's own function (tests/golden/rel_assign.npz) and against the oracle on larger seeded batches."""
import os

import numpy as np
import pytest
import torch

from oracle import sgg_oracle as O

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'rel_assign.npz'))
DEV = 'cuda:0'


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize('tag', [str(t) for t in G['cases']])
def test_rel_assignments_equals_reference(tag):
    from sgg_amd.rel_assignments import rel_assignments
    for k in range(5):
        per_gt, nonov, seed = [int(v) for v in G['%s_cfg%d' % (tag, k)]]
        np.random.seed(seed)
        out = rel_assignments(_t(G[tag + '_im_inds']), _t(G[tag + '_boxes']), _t(G[tag + '_labels']), _t(G[tag + '_gt_boxes']),
                              _t(G[tag + '_gt_classes']), _t(G[tag + '_gt_rels']), 0, num_sample_per_gt=per_gt,
                              filter_non_overlap=bool(nonov))
        assert out.is_cuda and out.dtype == torch.int64
        np.testing.assert_array_equal(out.cpu().numpy(), G['%s_out%d' % (tag, k)], err_msg='%s case %d' % (tag, k))


def test_rel_assign_tables_bit_exact_vs_oracle():
    """the tables themselves (fp32 IoU bits, match, candidates) on 8 images x ~50 detections"""
    from sgg_amd import ops
    rng = np.random.RandomState(3)
    B, ng, nd = 8, 20, 50
    xy = rng.uniform(0, 400, size=(B * ng, 2)); wh = rng.uniform(20, 190, size=(B * ng, 2))
    gt_boxes = np.concatenate((xy, xy + wh), 1).astype(np.float32)
    gt_classes = np.stack((np.repeat(np.arange(B), ng), rng.randint(1, 151, B * ng)), 1).astype(np.int64)
    pick = np.concatenate([i * ng + rng.randint(ng, size=nd) for i in range(B)])
    det = (gt_boxes[pick] + rng.uniform(-10, 10, size=(B * nd, 4))).astype(np.float32)
    det[::7] = gt_boxes[pick][::7]                      # exact copies: IoU == 1 with their GT and with each other
    lab = np.where(rng.rand(B * nd) < 0.8, gt_classes[pick, 1], rng.randint(0, 151, B * nd)).astype(np.int64)
    lab[::11] = 0
    im = np.repeat(np.arange(B), nd).astype(np.int64)
    for nonov in (True, False):
        iou, match, poss = ops.rel_assign_tables(_t(det), _t(im), _t(lab), _t(gt_boxes), _t(gt_classes), 0.5, nonov)
        ref_iou = O.box_iou(det, gt_boxes)
        same = im[:, None] == gt_classes[None, :, 0]
        np.testing.assert_array_equal(iou.cpu().numpy()[same].view(np.uint32), ref_iou[same].view(np.uint32))
        assert (iou.cpu().numpy()[~same] == -1).all()
        np.testing.assert_array_equal(match.cpu().numpy().astype(bool), same & (lab[:, None] == gt_classes[None, :, 1]) & (ref_iou >= 0.5))
        self_iou = O.box_iou(det, det)
        want = (im[:, None] == im[None]) & (lab[:, None] != 0) & (lab[None] != 0)
        want &= ((self_iou < 1) & (self_iou > 0)) if nonov else ~np.eye(len(det), dtype=bool)
        np.testing.assert_array_equal(poss.cpu().numpy().astype(bool), want)


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_rel_assignments_vs_oracle_larger(seed):
    from sgg_amd.rel_assignments import rel_assignments
    rng = np.random.RandomState(100 + seed)
    B, ng, nd = 6, 14, 45
    xy = rng.uniform(0, 380, size=(B * ng, 2)); wh = rng.uniform(30, 200, size=(B * ng, 2))
    gt_boxes = np.concatenate((xy, xy + wh), 1).astype(np.float32)
    gt_classes = np.stack((np.repeat(np.arange(B), ng), rng.randint(1, 151, B * ng)), 1).astype(np.int64)
    gt_rels = np.array([(i, a, b, rng.randint(1, 51)) for i in range(B) for a, b in
                        [rng.choice(ng, 2, replace=False) for _ in range(10)]], dtype=np.int64)
    pick = np.concatenate([i * ng + rng.randint(ng, size=nd) for i in range(B)])
    det = (gt_boxes[pick] + rng.uniform(-6, 6, size=(B * nd, 4))).astype(np.float32)
    lab = np.where(rng.rand(B * nd) < 0.85, gt_classes[pick, 1], 0).astype(np.int64)
    im = np.repeat(np.arange(B), nd).astype(np.int64)
    for per_gt in (1, 4):
        np.random.seed(seed)
        want = O.rel_assignments(im, det, lab, gt_boxes, gt_classes, gt_rels, 0, num_sample_per_gt=per_gt)
        np.random.seed(seed)
        got = rel_assignments(_t(im), _t(det), _t(lab), _t(gt_boxes), _t(gt_classes), _t(gt_rels), 0, num_sample_per_gt=per_gt)
        np.testing.assert_array_equal(got.cpu().numpy(), want)
        assert (want[:, 3] > 0).sum() > 0
