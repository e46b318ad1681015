"""Known answers for the oracle's restatement of torchvision's FPN detector pieces (PARITY UNPINNED [3P]: no reference fixture pins them, so
the published definitions are checked on values one can do by hand): LevelMapper (FPN paper eq. 1), the per-level anchors, the pyramid's
shapes and its top-down structure."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import sgg_oracle as O  # noqa: E402


def test_level_mapper_known_answers():
    side = torch.tensor([8., 111., 112., 223., 224., 447., 448., 2000.])
    boxes = torch.stack((torch.zeros(8), torch.zeros(8), side, side), 1)
    # k = floor(4 + log2(side / 224)) clamped to [2, 5], as an index from 0: 224 -> 4 -> 2; 112 -> 3 -> 1; 448 -> 5 -> 3
    assert O.fpn_level_of(boxes).tolist() == [0, 0, 1, 1, 2, 2, 3, 3]
    # non-square boxes go by sqrt(area)
    assert O.fpn_level_of(torch.tensor([[0., 0., 448., 112.]])).tolist() == [2]


def test_level_anchors_known_answers():
    a = O.fpn_level_anchors(32)
    assert a.tolist() == [[-23., -11., 23., 11.], [-16., -16., 16., 16.], [-11., -23., 11., 23.]]       # ratios 0.5, 1, 2 (h / w), rounded
    assert O.fpn_level_anchors(512).tolist() == [[-362., -181., 362., 181.], [-256., -256., 256., 256.], [-181., -362., 181., 362.]]


def test_pyramid_shapes_and_top_down_dependence():
    torch.manual_seed(0)
    p = {}
    cin = 64
    p['detector.backbone.body.conv1.weight'] = torch.randn(64, 3, 7, 7) * 0.05
    bn = lambda n, c: p.update({n + '.weight': torch.ones(c), n + '.bias': torch.zeros(c), n + '.running_mean': torch.zeros(c), n + '.running_var': torch.ones(c)})  # noqa: E731
    bn('detector.backbone.body.bn1', 64)
    for li, (mid, cout, blocks, stride) in enumerate(O.RESNET50_LAYERS):
        for b in range(blocks):
            n = 'detector.backbone.body.layer%d.%d.' % (li + 1, b)
            ci = cin if b == 0 else cout
            p[n + 'conv1.weight'] = torch.randn(mid, ci, 1, 1) * (2.0 / ci) ** 0.5
            p[n + 'conv2.weight'] = torch.randn(mid, mid, 3, 3) * (2.0 / (9 * mid)) ** 0.5
            p[n + 'conv3.weight'] = torch.randn(cout, mid, 1, 1) * (1.0 / mid) ** 0.5
            for k, c in (('bn1', mid), ('bn2', mid), ('bn3', cout)):
                bn(n + k, c)
            if b == 0:
                p[n + 'downsample.0.weight'] = torch.randn(cout, ci, 1, 1) * (1.0 / ci) ** 0.5
                bn(n + 'downsample.1', cout)
        cin = cout
    for k, (_, c, _, _) in enumerate(O.RESNET50_LAYERS):
        p['detector.backbone.fpn.inner_blocks.%d.weight' % k] = torch.randn(256, c, 1, 1) * (1.0 / c) ** 0.5
        p['detector.backbone.fpn.inner_blocks.%d.bias' % k] = torch.zeros(256)
        p['detector.backbone.fpn.layer_blocks.%d.weight' % k] = torch.randn(256, 256, 3, 3) * 0.02
        p['detector.backbone.fpn.layer_blocks.%d.bias' % k] = torch.zeros(256)
    x = torch.randn(1, 3, 64, 96)
    with torch.no_grad():
        lv = O.resnet50_fpn_levels(x, p)
        assert [tuple(t.shape[-2:]) for t in lv] == [(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)]
        assert torch.equal(lv[4], lv[3][:, :, ::2, ::2])                      # LastLevelMaxPool: kernel 1, stride 2
        assert torch.equal(O.resnet50_fpn_pool(x, p), lv[4])
        # the top-down pathway: changing the TOP lateral changes every level; changing the bottom lateral changes P2 only
        q = dict(p)
        q['detector.backbone.fpn.inner_blocks.3.bias'] = torch.ones(256)
        lt = O.resnet50_fpn_levels(x, q)
        assert all(not torch.equal(a, b) for a, b in zip(lv, lt))
        q = dict(p)
        q['detector.backbone.fpn.inner_blocks.0.bias'] = torch.ones(256)
        lb = O.resnet50_fpn_levels(x, q)
        assert not torch.equal(lv[0], lb[0]) and all(torch.equal(a, b) for a, b in zip(lv[1:], lb[1:]))
