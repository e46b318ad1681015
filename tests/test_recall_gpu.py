"""f-1: the recall evaluator on the HIP path (sgg_amd/recall.py, eval.hip) against the reference evaluator's golden
outputs (tests/golden/recall.npz, made by running lib/sgg_eval.py in the build container) and against the oracle."""
import contextlib
import io

import numpy as np
import pytest
import torch

from oracle import sgg_oracle as O

pytestmark = pytest.mark.gpu
KS = (20, 50, 100, 200, 300)
NO = 2 ** 31 - 1


@pytest.fixture(scope='module')
def R():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from sgg_amd import recall
    return recall


def cases_of(d):
    n = int(d['n_cases'])
    return [{k[len('c%d_' % i):]: d[k] for k in d if k.startswith('c%d_' % i)} for i in range(n)]


def split(c):
    return ({k: c[k] for k in ('gt_boxes', 'gt_classes', 'gt_relations')},
            {k: c[k].copy() for k in ('pred_boxes', 'pred_classes', 'obj_scores', 'pred_rel_inds', 'rel_scores')})


@pytest.mark.parametrize('mode', ['predcls', 'sgcls', 'sgdet', 'phrdet', 'objcls', 'preddet'])
@pytest.mark.parametrize('mp', [0, 1])
def test_evaluator_matches_reference_goldens(R, golden, mode, mp):
    d = golden('recall')
    ev = R.BasicSceneGraphEvaluator(mode, multiple_preds=bool(mp))
    evb = R.BasicSceneGraphEvaluator(mode, multiple_preds=bool(mp))
    cases = cases_of(d)
    firsts, nmatch = [], []
    for c in cases:
        gt, pr = split(c)
        pred_to_gt, five, _ = ev.evaluate_scene_graph_entry(gt, pr)
        if pred_to_gt is None:
            continue
        first = np.full(len(c['gt_relations']), NO, np.int64)
        for p, lst in enumerate(pred_to_gt):
            for g in lst:
                first[g] = min(first[g], p)
        firsts.append(first)
        nmatch.append(sum(len(x) for x in pred_to_gt))
        assert five.shape == (len(pred_to_gt), 5)
    evb.evaluate_scene_graph_batch(*zip(*[split(c) for c in cases]))           # one launch for all images
    for k in KS:
        exp = d['recall_%s_%d_%d' % (mode, mp, k)]
        np.testing.assert_allclose(ev.result_dict[mode + '_recall'][k], exp, rtol=0, atol=1e-12)
        np.testing.assert_allclose(evb.result_dict[mode + '_recall'][k], exp, rtol=0, atol=1e-12)
    if mode == 'preddet':
        return
    np.testing.assert_array_equal(nmatch, d['nmatch_%s_%d' % (mode, mp)])      # every match, not only the first
    got, exp = np.concatenate(firsts), d['first_%s_%d' % (mode, mp)]
    if (mode, mp) == ('objcls', 1):                                            # exact score ties: order unspecified upstream
        np.testing.assert_array_equal(got == NO, exp == NO)
    else:
        np.testing.assert_array_equal(got, exp)


def test_per_triplet_tables_and_mean_recall(R, golden):
    d = golden('recall')
    cases = cases_of(d)
    counts = dict(zip([str(k) for k in d['tc_keys']], [int(v) for v in d['tc_vals']]))
    quiet = contextlib.redirect_stdout(io.StringIO())
    for batch in (False, True):
        ev = R.BasicSceneGraphEvaluator('sgcls', multiple_preds=True, per_triplet=True, triplet_counts=counts)
        if batch:
            ev.evaluate_scene_graph_batch(*zip(*[split(c) for c in cases]))
        else:
            for c in cases:
                ev.evaluate_scene_graph_entry(*split(c))
        with quiet:
            ev.print_stats()
        rd = ev.result_dict
        np.testing.assert_array_equal(np.array(rd['sgcls_rank']), d['pt_rank'])
        np.testing.assert_array_equal(np.array(rd['sgcls_counts']), d['pt_counts'])
        for k in KS:
            np.testing.assert_allclose(rd['sgcls_recall_norm'][k], d['pt_recall_norm_%d' % k], rtol=1e-12, atol=1e-15)
        for sfx in ('', '_norm'):
            np.testing.assert_allclose([rd['sgcls_recall_triplet' + sfx][k] for k in (5, 10, 15, 20, 50)],
                                       d['pt_recall_triplet' + sfx], rtol=1e-6)
        got = [rd['sgcls_meanrank_triplet'], rd['sgcls_meanrank_triplet_norm'], rd['sgcls_medianrank_triplet'],
               rd['sgcls_medianrankclass_triplet'], rd['sgcls_medianrank_triplet_norm']]
        np.testing.assert_allclose(got, d['pt_scalars'], rtol=1e-6)
    with pytest.raises(NameError):          # the reference's per-triplet block reads overall_scores, defined only without GC
        R.BasicSceneGraphEvaluator('sgcls', per_triplet=True, triplet_counts=counts).evaluate_scene_graph_entry(*split(cases[0]))
    for mp in (0, 1):
        lst = [(pid, 'p%d' % pid, R.BasicSceneGraphEvaluator.all_modes(multiple_preds=bool(mp))) for pid in range(1, 9)]
        other = [(pid, 'p%d' % pid, R.BasicSceneGraphEvaluator.all_modes(multiple_preds=not mp)) for pid in range(1, 9)]
        for c in cases:
            gt, pr = split(c)
            R.eval_entry('sgcls', gt, pr, lst if not mp else other, other if not mp else lst)
        with quiet:
            mr = R.calculate_mR_from_evaluator_list(lst, 'sgcls', multiple_preds=bool(mp))
        np.testing.assert_allclose([mr['R@%d' % k] for k in KS], d['mR_%d' % mp], rtol=1e-12)


@pytest.mark.parametrize('phrdet', [False, True])
def test_first_match_kernel_vs_oracle_large(R, phrdet):
    """Bigger than anything the fixtures hold: 3 images, up to 5000 ranked predictions each, boxes that straddle IoU 0.5."""
    rng = np.random.RandomState(3 + phrdet)
    cases, expect = [], []
    for img in range(3):
        nb, G, P = 24, 40 + 7 * img, (5000, 64, 1300)[img]
        boxes = np.concatenate((rng.uniform(0, 300, (nb, 2)), rng.uniform(300, 500, (nb, 2))), 1).astype(np.float32)
        pboxes = (boxes + rng.uniform(-45, 45, boxes.shape)).astype(np.float32)
        gcls, pcls = rng.randint(1, 5, nb), None
        pcls = gcls.copy()
        pcls[rng.rand(nb) < 0.2] = 1
        gt_rels = np.column_stack((rng.randint(0, nb, G), rng.randint(0, nb, G), rng.randint(1, 4, G)))
        pred_rels = np.column_stack((rng.randint(0, nb, P), rng.randint(0, nb, P), rng.randint(1, 4, P)))
        cases.append(R._make_case(gt_rels, boxes, gcls, pred_rels, pboxes, pcls))
        p2g = O.recall_pred_to_gt(gt_rels, boxes, gcls, pred_rels, pboxes, pcls, 0.5, phrdet)
        first = np.full(G, NO, np.int64)
        for p, lst in enumerate(p2g):
            for g in lst:
                first[g] = min(first[g], p)
        expect.append(first)
    got = R.match_cases(cases, 0.5, phrdet=phrdet)
    for (first, _), exp in zip(got, expect):
        np.testing.assert_array_equal(first, exp)
    assert any((e != NO).any() for e in expect) and any((e == NO).any() for e in expect)


def test_recall_of_model_output_matches_oracle_evaluator(R):
    """R@K parity end to end on synthetic data: the HIP forward's eval tuple through the HIP evaluator equals the oracle
    evaluator applied to the same tuple (sgcls and predcls, GC and no-GC)."""
    import sgg_amd
    from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
    S = 128
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls', min_size=S, max_size=S)).to('cuda:0').eval()
    model.set_compute_dtype(torch.float32)
    batch = synthetic_batch(B=1, S=S, n_boxes=9, n_fg=5, seed=21)
    for mode in ('sgcls', 'predcls'):
        model.mode = mode
        with torch.no_grad():
            boxes, cls, scores, rels, pred_scores = model([batch])
        gt_rels = batch[5][:, 1:].numpy()
        gt = {'gt_boxes': batch[3].numpy(), 'gt_classes': batch[4][:, 1].numpy(), 'gt_relations': gt_rels}
        pr = {'pred_boxes': boxes, 'pred_classes': cls, 'obj_scores': scores, 'pred_rel_inds': rels, 'rel_scores': pred_scores}
        for mp in (False, True):
            ev = R.BasicSceneGraphEvaluator(mode, multiple_preds=mp)
            ev.evaluate_scene_graph_entry(gt, pr)
            rec, _, _ = O.recall_entry(gt, pr, mode, mp)
            for k in KS:
                assert ev.result_dict[mode + '_recall'][k][0] == rec[k]
