"""GPU parity of the SGDet front end (SURVEY a-12, BASELINE config 3): RPN + RoI heads + post-processing on the HIP
path vs the CPU oracle's restatement of torchvision's eval-mode FasterRCNN (PARITY UNPINNED [3P]), then the whole
SGDet forward.  fp32 mode; score threshold 0 so that random-init weights produce detections."""
import numpy as np
import pytest
import torch

from oracle import sgg_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def env():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import sgg_amd
    from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
    S = 128
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgdet', min_size=S, max_size=S))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(DEV).eval().set_compute_dtype(torch.float32)
    model.set_box_score_thresh(0.0)
    batch = synthetic_batch(B=2, S=S, n_boxes=4, n_fg=2, seed=2)
    with torch.no_grad():
        ref = O.forward_sgdet(batch[0], sd, score_thresh=0.0, min_size=S, max_size=S)
    return model, sd, batch, ref, S


def test_nms_and_sort_kernels_vs_oracle(env):
    from sgg_amd import sgdet
    g = torch.Generator().manual_seed(3)
    B, n = 3, 700
    xy = torch.rand(B, n, 2, generator=g) * 300
    wh = torch.rand(B, n, 2, generator=g) * 120 + 5
    boxes = torch.cat((xy, xy + wh), 2)
    scores = torch.rand(B, n, generator=g)
    labels = torch.randint(1, 6, (B, n), generator=g, dtype=torch.int32)
    seg = torch.arange(0, B + 1, dtype=torch.int32) * n
    ks, vs = sgdet._sort_desc(scores.view(-1).to(DEV), seg.to(DEV), B, n)
    for b in range(B):
        exp_s, exp_i = torch.sort(scores[b], descending=True, stable=True)
        torch.testing.assert_close(ks.view(B, n)[b].cpu(), exp_s)
        np.testing.assert_array_equal(vs.view(B, n)[b].cpu().numpy(), exp_i.numpy())
    order = vs.view(B, n).long().cpu()
    sb = torch.stack([boxes[b][order[b]] for b in range(B)]).to(DEV).contiguous()
    sl = torch.stack([labels[b][order[b]] for b in range(B)]).to(DEV).contiguous()
    valid = torch.ones(B, n, dtype=torch.uint8, device=DEV)
    for lab, thr, cap in ((None, 0.7, 1000), (sl, 0.5, 50)):
        kidx, kcnt = sgdet._nms(sb, lab, valid, thr, min(cap, n))
        for b in range(B):
            bb = sb[b].cpu()
            off = (sl[b].cpu().float() * 1000.0)[:, None] if lab is not None else 0
            exp = O.nms(bb + off, torch.arange(n, 0, -1).float(), thr)[:cap]
            got = kidx[b, :int(kcnt[b])].cpu().long()
            np.testing.assert_array_equal(got.numpy(), exp.numpy())


def test_detections_match_oracle(env):
    model, sd, batch, ref, S = env
    with torch.no_grad():
        res = model.faster_rcnn(batch[0], None, batch[4].to(DEV), None)
    exp_b = np.concatenate([d[0].numpy() for d in ref['detections']])
    exp_l = np.concatenate([d[2].numpy() for d in ref['detections']])
    got_b, got_l = res.rm_box_priors.cpu().numpy(), res.rm_obj_labels.cpu().numpy()
    assert got_b.shape == exp_b.shape, (got_b.shape, exp_b.shape)
    np.testing.assert_array_equal(got_l, exp_l)
    np.testing.assert_allclose(got_b, exp_b, atol=2e-3)
    np.testing.assert_array_equal(res.im_inds.cpu().numpy(), np.repeat(np.arange(2), [len(d[0]) for d in ref['detections']]))
    assert max(len(d[0]) for d in ref['detections']) <= 50


def test_sgdet_forward_matches_oracle(env):
    model, sd, batch, ref, S = env
    with torch.no_grad():
        boxes, cls, scores, rels, pred_scores = model([batch])
    rb, rc, rs, rr, rp = ref['dets']
    np.testing.assert_allclose(boxes, rb, atol=2e-3)
    np.testing.assert_array_equal(cls, rc)
    np.testing.assert_allclose(scores, rs, atol=1e-3)
    assert rels.shape == rr.shape                       # overlap-filtered pair list
    np.testing.assert_array_equal(rels, rr)
    np.testing.assert_allclose(pred_scores, rp, atol=1e-3)


def test_sgdet_too_few_detections_raises_value_error(env):
    model, sd, batch, ref, S = env
    model.set_box_score_thresh(0.999)                  # nothing passes: eval.py:227 catches this and retries lower
    with pytest.raises(ValueError):
        with torch.no_grad():
            model([batch])
    model.set_box_score_thresh(0.0)
    model.train()
    with pytest.raises(NotImplementedError):
        model([batch])
    model.eval()
