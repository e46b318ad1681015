"""Evaluation harness (lib/eval.py val_epoch / val_batch, lib/pytorch_misc.py set_mode) over the HIP forward and the HIP recall
matcher: the logged R@K equal the oracle evaluator applied to the same per-image predictions."""
import contextlib
import io

import numpy as np
import pytest
import torch

from oracle import sgg_oracle as O

pytestmark = pytest.mark.gpu
KS = (20, 50, 100)
S = 128


class _Split(object):
    """What val_epoch / val_batch read from a VG split (dataloaders/visual_genome.py:377-455): BOX_SCALE-space gt lists."""
    torch_detector = True
    split = 'stanford'

    def __init__(self, n_img=4, seed=31):
        from sgg_amd.synthetic import SyntheticData, synthetic_batch
        self.ind_to_classes, self.ind_to_predicates = SyntheticData.ind_to_classes, SyntheticData.ind_to_predicates
        self.num_classes, self.num_predicates = 151, 51
        self.data, self.gt_classes, self.relationships, self.gt_boxes = [], [], [], []
        for i in range(n_img):
            imgs, _, _, boxes, cls, rels, _, fns = synthetic_batch(B=1, S=S, n_boxes=6 + i, n_fg=4, seed=seed + i)
            self.data.append({'img': imgs[0], 'img_size': (S, S, 1.0), 'gt_boxes': boxes.numpy(), 'gt_classes': cls[:, 1].numpy(),
                              'gt_relations': rels[:, 1:].numpy(), 'scale': 1.0, 'fn': fns[0]})
            self.gt_classes.append(cls[:, 1].numpy())
            self.relationships.append(rels[:, 1:].numpy())
            self.gt_boxes.append(boxes.numpy() * (1024.0 / S))

    def __len__(self):
        return len(self.data)


class _Loader(object):
    def __init__(self, dataset):
        self.dataset = dataset

    def __iter__(self):
        from sgg_amd.blob import vg_collate
        for d in self.dataset.data:
            yield vg_collate([d], mode='rel', is_train=False)


@pytest.fixture(scope='module')
def model():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import sgg_amd
    from sgg_amd.synthetic import SyntheticData, init_weights
    m = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls', min_size=S, max_size=S)).to('cuda:0').eval()
    m.set_compute_dtype(torch.float32)
    return m


def test_set_mode(model):
    from sgg_amd.evaluate import set_mode
    set_mode(model, 'sgdet', is_train=False)
    assert model.mode == 'sgdet' and model.detector.mode == 'refinerels' and not model.training
    set_mode(model, 'predcls', is_train=True)
    assert model.mode == 'predcls' and model.detector.mode == 'gtbox' and model.training
    set_mode(model, 'sgcls', is_train=False)


def test_val_epoch_recalls_equal_oracle_evaluator(model):
    from sgg_amd.evaluate import val_epoch
    ds = _Split()
    results, logged = {}, []
    with contextlib.redirect_stdout(io.StringIO()):
        entries = val_epoch('sgcls', model, _Loader(ds), 'test', None, None, is_test=True, save_scores=True, results=results,
                            wandb_log=lambda d, **kw: logged.append((d, kw)))
    assert set(entries) == {'predcls', 'sgcls'} and all(len(v) == len(ds) for v in entries.values())
    assert logged and logged[0][1]['log_repeats'] == 5 and logged[0][1]['is_summary']
    for mode in ('predcls', 'sgcls'):
        for mp, sfx in ((False, 'GC'), (True, 'NOGC')):
            per_k = {k: [] for k in KS}
            for i, pr in enumerate(entries[mode]):
                gt = {'gt_boxes': ds.gt_boxes[i] * (S / 1024.0), 'gt_classes': ds.gt_classes[i], 'gt_relations': ds.relationships[i]}
                rec, _, _ = O.recall_entry(gt, pr, mode, mp)
                for k in KS:
                    per_k[k].append(rec[k])
            for k in KS:
                assert results['%s/test_R@%d_%s' % (mode, k, sfx)] == pytest.approx(np.mean(per_k[k]), abs=1e-12)
            assert np.isfinite(results['%s/test_mR@50_%s' % (mode, sfx)])
    assert 'avg/test_R' in results
    # a validation split name skips the per-predicate evaluators (lib/eval.py:45) and n_batches stops early
    r2 = {}
    with contextlib.redirect_stdout(io.StringIO()):
        e2 = val_epoch('sgcls', model, _Loader(ds), 'val_zs', None, None, n_batches=2, save_scores=True, results=r2)
    assert len(e2['sgcls']) == 2 and not any('_mR@' in k for k in r2)


def test_val_batch_predicate_reweighting(model):
    """lib/eval.py:163-167: rel_scores[:, 1:] / weights, renormalised"""
    from sgg_amd.evaluate import predicate_weights_from, set_mode, val_batch
    from sgg_amd.recall import BasicSceneGraphEvaluator
    ds = _Split(n_img=2)
    w = predicate_weights_from(ds, 0.7)
    assert w.shape == (51,) and (w > 0).all()
    set_mode(model, 'sgcls', is_train=False)
    batch = next(iter(_Loader(ds)))
    ev = lambda: {'sgcls': BasicSceneGraphEvaluator('sgcls'), 'sgcls_nogc': BasicSceneGraphEvaluator('sgcls', multiple_preds=True)}
    with torch.no_grad():
        plain = val_batch(model, 0, batch, ev(), 'sgcls', ds, [], [])[0]
        rew = val_batch(model, 0, batch, ev(), 'sgcls', ds, [], [], predicate_weights=w)[0]
    want = plain['rel_scores'].copy()
    want[:, 1:] = want[:, 1:] / w[1:]
    want = want / want.sum(1, keepdims=True)
    np.testing.assert_allclose(rew['rel_scores'], want, rtol=1e-6)
    with pytest.raises(NotImplementedError):
        val_batch(model, 0, batch, ev(), 'sgcls', ds, [], [], vis=True)


def test_val_epoch_over_vg_tables(model):
    """f-2 + path + f-1 chained: VG-SGG tables -> sgg_amd.visual_genome.VG -> vg_collate -> HIP forward (u8 images, SquarePad on the GPU)
    -> val_epoch; recalls equal the oracle evaluator on the same predictions and box scales (lib/eval.py:143-153)."""
    import os
    from sgg_amd.blob import vg_collate
    from sgg_amd.evaluate import val_epoch
    from sgg_amd.visual_genome import VG
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'vg_loader.npz'))
    tables = {k[3:]: g[k] for k in g.files if k.startswith('h5_')}
    rng = np.random.RandomState(2)
    cache = {}
    decode = lambda path: cache.setdefault(path, rng.randint(0, 255, size=(96, S, 3)).astype(np.uint8))
    info = {'label_to_idx': {'c%d' % i: i for i in range(1, 151)}, 'predicate_to_idx': {'p%d' % i: i for i in range(1, 51)}}
    ds = VG('test', tables, info, ['%d.jpg' % i for i in range(len(tables['split']))], num_val_im=0, decode=decode)

    class Loader(object):
        dataset = ds

        def __iter__(self):
            for i in range(len(ds)):
                yield vg_collate([ds[i]], mode='rel', is_train=False)
    results = {}
    with contextlib.redirect_stdout(io.StringIO()):
        entries = val_epoch('sgcls', model, Loader(), 'val_zs', None, None, save_scores=True, results=results)
    assert len(entries['sgcls']) == len(ds) > 3
    for mode in ('predcls', 'sgcls'):
        per_k = {k: [] for k in KS}
        for i, pr in enumerate(entries[mode]):
            gt = {'gt_boxes': ds.gt_boxes[i] * (S / 1024.0), 'gt_classes': ds.gt_classes[i], 'gt_relations': ds.relationships[i]}
            rec, _, _ = O.recall_entry(gt, pr, mode, False)
            for k in KS:
                per_k[k].append(rec[k])
        for k in KS:
            assert results['%s/val_zs_R@%d_GC' % (mode, k)] == pytest.approx(np.mean(per_k[k]), abs=1e-12)
    # the two box scales agree (BOX_SCALE tables -> image scale on both sides): predicted boxes ARE the clipped GT boxes
    hit = []
    for i, pr in enumerate(entries['predcls']):
        iou = O.box_iou(pr['pred_boxes'], ds.gt_boxes[i] * (S / 1024.0))
        hit.extend(np.diag(iou) > 0.5)
        np.testing.assert_array_equal(pr['pred_classes'], ds.gt_classes[i])
    assert np.mean(hit) > 0.8
