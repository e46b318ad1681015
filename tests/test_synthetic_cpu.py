"""Host-side pieces added in round 6 (no GPU): the synthetic detector that BEHAVES like a trained one (BASELINE configs[2] needs 1 000 proposals
per image: checked on the oracle's RPN stage), GQA's vocabulary stand-in, the parity clause a compute mode advertises."""
import numpy as np
import torch

from oracle import sgg_oracle as O


def test_spread_detector_gives_1000_proposals_and_50_detections_per_image_in_the_oracle():
    import sgg_amd
    from sgg_amd.synthetic import SyntheticData, init_weights, spread_detector_, synthetic_batch
    torch.set_num_threads(8)
    S = 592
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgdet'))
    plain = {k: v.clone() for k, v in model.state_dict().items()}
    sd = spread_detector_({k: v.clone() for k, v in plain.items()})
    batch = synthetic_batch(B=1, S=S, n_boxes=8, n_fg=2, seed=41)
    with torch.no_grad():
        x, sizes, _ = O.transform(batch[0], None, S, S)
        fmap = O.vgg16_features(x, sd)
        padded = tuple(x.shape[-2:])
        props = O.rpn_proposals(fmap, sd, sizes, padded)
        props_plain = O.rpn_proposals(fmap, plain, sizes, padded)
    assert [len(p) for p in props] == [1000]
    assert len(props_plain[0]) < 200            # the He-initialised RPN collapses under NMS 0.7: why the synthetic detector exists
    side = (props[0][:, 2:] - props[0][:, :2])
    assert float(side.max()) < 64.0             # the 32-px anchors (three ratios), barely moved
    with torch.no_grad():
        dets = O.roi_heads_detect(fmap, props, sd, sizes, 1.0 / 16, 0.05)
    bx, sc, lb = dets[0]
    assert len(bx) == 50 and float(sc.min()) > 0.5 and len(torch.unique(sc)) == 50      # distinct, confident scores
    iou = O.box_iou(bx.numpy(), bx.numpy())
    assert int((iou > 0).sum() - len(bx)) >= 200                                         # overlapping detections: candidate edges exist


def test_gqa_stand_in_and_relabel_batch():
    from sgg_amd.synthetic import GQASyntheticData, relabel_batch, synthetic_batch
    d = GQASyntheticData()
    assert len(d.ind_to_classes) == 1704 and len(d.ind_to_predicates) == 311 and d.ind_to_classes[0] == '__background__'
    b = synthetic_batch(B=2, S=64, n_boxes=5, n_fg=3, seed=1)
    r = relabel_batch(b, 1704, 311, seed=3)
    assert torch.equal(r[3], b[3]) and torch.equal(r[4][:, 0], b[4][:, 0]) and torch.equal(r[5][:, :3], b[5][:, :3])
    assert int(r[4][:, 1].min()) >= 1 and int(r[4][:, 1].max()) < 1704 and int(r[5][:, 3].min()) >= 1 and int(r[5][:, 3].max()) < 311
    assert torch.equal(b[4][:, 1], synthetic_batch(B=2, S=64, n_boxes=5, n_fg=3, seed=1)[4][:, 1])          # the source batch is untouched


def test_parity_clause_of_every_compute_mode(caplog):
    import logging
    import sgg_amd
    from sgg_amd import rel_model_base
    from sgg_amd.synthetic import SyntheticData
    rel_model_base._PARITY_SAID.clear()
    with caplog.at_level(logging.WARNING, logger='sgg_amd'):
        m = sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls', min_size=64, max_size=64)
        assert m.parity_clause == dict(m.parity_clause, mode='f16') and not m.parity_clause['logits_within_1e-3'] and m.parity_clause['recall_within_0.1']
        m.set_compute_dtype(torch.bfloat16)
        assert not m.parity_clause['logits_within_1e-3'] and not m.parity_clause['recall_within_0.1']
        m.set_compute_dtype(torch.float32, split3=True)
        assert m.parity_clause['mode'] == 'x3' and m.parity_clause['logits_within_1e-3'] and m.parity_clause['recall_within_0.1']
        m.set_compute_dtype(torch.float32)
        assert m.parity_clause['mode'] == 'f32' and m.parity_clause['logits_within_1e-3']
        m.set_compute_dtype(torch.float16)
    said = [r.getMessage() for r in caplog.records if 'compute mode' in r.getMessage()]
    assert len(said) == 2 and 'NOT met' in said[0] and 'bf16' in said[1]           # once per mode, only for the modes that miss a clause
    m.set_compute_dtype(torch.float16)
    import pytest
    with pytest.raises(ValueError):
        m.set_compute_dtype(torch.float32, backward_f16=True)
