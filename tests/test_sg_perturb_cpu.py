"""Scene-graph perturbations (augment/sg_perturb.py; main.py:131-134) against the reference's outputs under the same seeds
(tests/golden/sg_perturb.npz), and the dataset statistics they consult (dataloaders/visual_genome.py:211-227)."""
import os

import numpy as np
import pytest
import torch

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'sg_perturb.npz'))
CASES = [('rand', dict(L=0.5)), ('neigh', dict(L=0.3, topk=3)), ('graphn', dict(L=0.5, topk=0, alpha=2)),
         ('graphn', dict(L=1.0, topk=4, alpha=1)), ('rand', dict(L=0.2, uniform=True)),
         ('graphn', dict(L=0.4, topk=2, alpha=5, degree_smoothing=0.5))]


def stats():
    subj_pred, pred_obj = {}, {}
    for s_, p_, o_, cnt in G['stats_rows'].tolist():
        subj_pred.setdefault('{}_{}'.format(s_, p_), {})[o_] = cnt
        pred_obj.setdefault('{}_{}'.format(p_, o_), {})[s_] = cnt
    return subj_pred, pred_obj


@pytest.mark.parametrize('k', range(len(CASES)))
def test_perturbed_graphs_equal_reference_under_the_same_seeds(k):
    from sgg_amd.sg_perturb import SceneGraphPerturb
    method, kw = CASES[k]
    sgp = SceneGraphPerturb(method=method, embed_objs=torch.from_numpy(G['embed']).clone(), subj_pred_obj_pairs=stats(), **kw)
    gt_obj, gt_rels = torch.from_numpy(G['gt_obj']), torch.from_numpy(G['gt_rels'])
    changed = 0
    for rep in range(3):
        np.random.seed(100 * k + rep)
        torch.manual_seed(100 * k + rep)
        out = sgp.perturb(gt_obj.clone(), gt_rels.clone())
        want = torch.from_numpy(G['case%d_rep%d' % (k, rep)])
        assert torch.equal(out, want), (method, kw, rep)
        assert torch.equal(out[:, 0], gt_obj[:, 0]) and (out[:, 1] > 0).all()
        changed += int((out[:, 1] != gt_obj[:, 1]).sum())
    assert changed > 0


def test_constructor_rules_and_no_perturbation():
    from sgg_amd.sg_perturb import SceneGraphPerturb, pairwise_similarity
    emb = torch.from_numpy(G['embed'])
    with pytest.raises(ValueError):
        SceneGraphPerturb('neigh', None, stats(), topk=3)                 # no word vectors
    with pytest.raises(NotImplementedError):
        SceneGraphPerturb('swap', emb, stats())
    sim = pairwise_similarity(emb)
    assert torch.isinf(sim[0]).all() and torch.isinf(sim[:, 0]).all() and torch.isinf(sim.diagonal()).all()
    sgp = SceneGraphPerturb('rand', None, stats(), L=0.0, obj_classes=['c%d' % i for i in range(12)])   # L = 0: nothing is sampled
    gt_obj, gt_rels = torch.from_numpy(G['gt_obj']), torch.from_numpy(G['gt_rels'])
    assert torch.equal(sgp.perturb(gt_obj.clone(), gt_rels.clone()), gt_obj)
