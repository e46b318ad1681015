"""Evaluation harness around the path: what lib/eval.py (val_epoch / val_batch) and lib/pytorch_misc.py (set_mode)
provide to main.py, rebuilt around the batched HIP recall matcher.

The reference evaluates image by image: forward, then ~100 numpy evaluator calls per image (GC, no-GC and 2 x 50 per-predicate
evaluators).  Here the forward still runs one image per call (the Blob of dataloaders/visual_genome.py:730 holds one image per
GPU), but recall accounting is DEFERRED: a `SplitScorer` collects the (gt_entry, pred_entry) pairs of a split and scores them in
chunks -- one `sgg_recall_first_match` launch per evaluator and chunk instead of one per evaluator and image, and a predicate's
evaluators only ever see the images that contain that predicate.  The numbers are the reference's: same entries, same
evaluator semantics (sgg_amd/recall.py), same result keys.

Public surface (signatures as in the reference, so main.py calls it unchanged):
    set_mode(sgg_model, mode, is_train, verbose=False)
    val_batch(sgg_model, batch_num, b, evaluator, eval_m, val_dataset, evaluator_list, evaluator_multiple_preds_list, ...)
    val_epoch(mode, sgg_model, loader, name, triplet_counts, triplet2str, n_batches=-1, is_test=False, save_scores=False,
              predicate_weight=0, train=None, wandb_log=None, results=None)
Not carried over: the matplotlib / networkx drawing branch (`vis=True`) and wandb itself -- `wandb_log` may be any callable taking
(dict, step=..., is_summary=..., log_repeats=...)."""
import numpy as np
import torch

from .recall import RECALL_KS, BasicSceneGraphEvaluator, PredicateRecall
from .sparse_targets import get_counts

IM_SCALE = 592      # config.py:31
BOX_SCALE = 1024    # config.py:30
all_shot_splits = ['val_alls', 'test_alls']          # per-triplet tables are kept for these splits (lib/eval.py:12)
SCORE_CHUNK = 64    # images per matching launch


def set_mode(sgg_model, mode, is_train, verbose=False):
    """lib/pytorch_misc.py:76-95: train / eval switch plus the mode attributes of the model and its sub-modules."""
    sgg_model.train(bool(is_train))
    sgg_model.mode = mode
    for attr, value in (('detector', 'refinerels' if mode == 'sgdet' else 'gtbox'), ('context', mode)):
        sub = getattr(sgg_model, attr, None)
        if sub is not None:
            if verbose:
                print('setting %s mode for %s' % (value, attr))
            sub.mode = value


def predicate_weights_from(train, predicate_weight):
    """Predicate re-weighting vector of lib/eval.py:25-30: (mean over class pairs of the smoothed predicate counts) ** weight,
    the background slot holding the smoothed background count."""
    fg, bg = get_counts(train, must_overlap=True)
    counts = fg.astype(np.float64) + 1.0
    counts[:, :, 0] = bg + 2.0                       # (bg + 1) + 1, as the reference's two in-place steps leave it
    return counts.mean(axis=(0, 1)) ** predicate_weight


# ------------------------------------------------------------------------------------------------------ one image -> entries
def _box_thresholds(val_dataset):
    """Detector score thresholds to try in turn (lib/eval.py:125-129) and the prediction box scale."""
    if val_dataset.torch_detector:
        return [0.2, 0.05, 0.01], 1.0
    return [None], BOX_SCALE / IM_SCALE


def _gt_scale(val_dataset, blob_item):
    """GT boxes are stored at BOX_SCALE; the 'stanford' split brings them to the image's own size (lib/eval.py:143-147)."""
    if getattr(val_dataset, 'split', 'stanford') != 'stanford':
        return 1.0
    w, h = blob_item[1][0, :2]
    return float(max(w, h)) / BOX_SCALE


def _reweight(rel_scores, predicate_weights):
    """lib/eval.py:163-167: foreground predicate scores divided by their weight, rows renormalised."""
    out = np.array(rel_scores, copy=True)
    out[:, 1:] /= predicate_weights[1:]
    out /= out.sum(axis=1, keepdims=True)
    assert (np.abs(out.sum(1) - 1) < 1e-5).all(), out.sum(1)
    return out


def forward_entries(sgg_model, batch_num, b, val_dataset, predicate_weights=None):
    """One Blob through the model -> [(gt_entry, pred_entry)] (one pair per image of the Blob), or None when no detector
    threshold produced a usable image (the reference's retry loop, lib/eval.py:131-134,223-227)."""
    thresholds, pred_scale = _box_thresholds(val_dataset)
    for thresh in thresholds:
        sgg_model.set_box_score_thresh(thresh)
        try:
            dets = [sgg_model(b.scatter())]
        except (ValueError, IndexError) as err:
            print('NO OBJECTS OR RELATIONS FOUND', err, b[0][-1], 'trying a smaller threshold')
            continue
        pairs = []
        for i, (boxes, classes, obj_scores, rels, rel_scores) in enumerate(dets):
            k = batch_num + i
            gt_entry = {'gt_classes': val_dataset.gt_classes[k].copy(),
                        'gt_relations': val_dataset.relationships[k].copy(),
                        'gt_boxes': val_dataset.gt_boxes[k].copy() * _gt_scale(val_dataset, b[i])}
            if predicate_weights is not None:
                rel_scores = _reweight(rel_scores, predicate_weights)
            pairs.append((gt_entry, {'pred_boxes': boxes * pred_scale, 'pred_classes': classes, 'pred_rel_inds': rels,
                                     'obj_scores': obj_scores, 'rel_scores': rel_scores}))
        return pairs
    return None


# ------------------------------------------------------------------------------------------------------ deferred scoring
class SplitScorer(object):
    """All recall tables of one (split, eval mode): graph-constrained, unconstrained (+ per-triplet tables on the all-shot
    splits) and -- for the splits the reference computes mean recall on -- one pair of evaluators per predicate.
    `add` queues an image; scoring happens in chunks of SCORE_CHUNK images."""

    def __init__(self, eval_m, name, ind_to_predicates, triplet_counts, triplet2str):
        self.eval_m, self.name = eval_m, name
        self.gc = BasicSceneGraphEvaluator(eval_m)
        self.nogc = BasicSceneGraphEvaluator(eval_m, multiple_preds=True, per_triplet=name in all_shot_splits,
                                             triplet_counts=triplet_counts, triplet2str=triplet2str)
        per_predicate = name not in ('val_zs', 'test_zs') and name.find('val_') < 0      # lib/eval.py:47
        self.by_predicate = PredicateRecall(ind_to_predicates) if per_predicate else None
        self._queue = []

    def add(self, gt_entry, pred_entry):
        self._queue.append((gt_entry, pred_entry))
        if len(self._queue) >= SCORE_CHUNK:
            self.flush()

    def flush(self):
        if not self._queue:
            return
        gts, preds = [g for g, _ in self._queue], [p for _, p in self._queue]
        self._queue = []
        self.gc.evaluate_scene_graph_batch(gts, preds)
        self.nogc.evaluate_scene_graph_batch(gts, preds)
        if self.by_predicate is not None:
            self.by_predicate.evaluate_batch(self.eval_m, gts, preds)

    def report(self):
        """Prints the tables and returns {result key: value} with the reference's key layout (lib/eval.py:84-110)."""
        self.flush()
        self.gc.print_stats()
        self.nogc.print_stats()
        mean_recall = {'GC': None, 'NOGC': None}
        if self.by_predicate is not None:
            mean_recall['GC'] = self.by_predicate.mean_recall(self.eval_m, multiple_preds=False)
            mean_recall['NOGC'] = self.by_predicate.mean_recall(self.eval_m, multiple_preds=True)
        m, name = self.eval_m, self.name
        out, image_means = {}, []
        for tag, ev in (('GC', self.gc), ('NOGC', self.nogc)):
            for k, per_image in ev.result_dict[m + '_recall'].items():
                r = float(np.mean(per_image))
                image_means.append(r)
                out['%s/%s_R@%i_%s' % (m, name, k, tag)] = r
            for key, val in (mean_recall[tag] or {}).items():
                out['%s/%s_m%s_%s' % (m, name, key, tag)] = float(np.mean(val))
        if name in all_shot_splits:
            rd = self.nogc.result_dict
            try:
                for case in ('', '_norm'):
                    for k, val in rd[m + '_recall_triplet' + case].items():
                        out['%s/%s_R@%i_triplet%s' % (m, name, k, case)] = val
                    for metric in ('meanrank', 'medianrank') + (('medianrankclass',) if case == '' else ()):
                        out['%s/%s_%s_triplet%s' % (m, name, metric, case)] = rd['%s_%s_triplet%s' % (m, metric, case)]
            except Exception as err:
                print('error in per triplet eval', err)
        return out, image_means


def val_batch(sgg_model, batch_num, b, evaluator, eval_m, val_dataset, evaluator_list, evaluator_multiple_preds_list,
              vis=False, max_obj=10, max_rels=20, train=None, test_zs=None, predicate_weights=None):
    """The reference's per-batch entry point (lib/eval.py:120-227), for callers that drive their own evaluators: forwards the
    Blob, feeds `evaluator[eval_m]` / `evaluator[eval_m + '_nogc']` and the per-predicate evaluator lists immediately, and
    returns the list of pred_entry dicts (None if no detector threshold gave boxes).  `val_epoch` does not go through here: it
    scores in chunks."""
    if vis:
        raise NotImplementedError('vis=True (matplotlib / networkx drawing, lib/eval.py:180-221) is outside the path')
    pairs = forward_entries(sgg_model, batch_num, b, val_dataset, predicate_weights)
    if pairs is None:
        return None
    gts, preds = [g for g, _ in pairs], [p for _, p in pairs]
    for sfx in ('', '_nogc'):
        evaluator[eval_m + sfx].evaluate_scene_graph_batch(gts, preds)
    if evaluator_list:
        PredicateRecall.from_lists(evaluator_list, evaluator_multiple_preds_list).evaluate_batch(eval_m, gts, preds)
    return preds


def val_epoch(mode, sgg_model, loader, name, triplet_counts, triplet2str, n_batches=-1, is_test=False, save_scores=False,
              predicate_weight=0, train=None, wandb_log=None, results=None, **kwargs):
    """lib/eval.py:15-117: every evaluation mode of `mode`'s family over the split `name`; returns {eval mode: [pred_entry]}
    (filled when save_scores).  `results` (optional dict, not in the reference) receives the keys the reference hands to wandb
    ('sgcls/test_R@50_GC', 'sgcls/test_mR@50_NOGC', 'avg/test_R', ...) whether or not `wandb_log` is given."""
    if kwargs.get('vis'):
        raise NotImplementedError('vis=True (matplotlib / networkx drawing, lib/eval.py:180-221) is outside the path')
    family = ['sgdet'] if mode == 'sgdet' else ['predcls', 'sgcls']
    assert mode in family, (mode, 'other modes not supported')
    print('\nEvaluate %s %s triplets' % (name.upper(), 'test' if is_test else 'val'))
    weights = predicate_weights_from(train, predicate_weight) if predicate_weight != 0 else None
    step = getattr(sgg_model, 'global_batch_iter', 0)
    want_metrics = bool(wandb_log) or results is not None

    def publish(d):
        if results is not None:
            results.update(d)
        if wandb_log:
            wandb_log(d, step=step, is_summary=True, log_repeats=5 if is_test else 1)

    saved, grand = {}, []
    sgg_model.eval()
    with torch.no_grad():
        for eval_m in family:
            if eval_m == 'sgdet' and 'val_' in name:
                continue                                  # the reference skips SGDet on validation splits (too slow there)
            print('\nEvaluating %s...' % eval_m.upper())
            set_mode(sgg_model, mode=eval_m, is_train=False, verbose=True)
            scorer = SplitScorer(eval_m, name, loader.dataset.ind_to_predicates, triplet_counts, triplet2str)
            saved[eval_m] = []
            for val_b, blob in enumerate(loader):
                pairs = forward_entries(sgg_model, val_b, blob, loader.dataset, weights)
                for gt_entry, pred_entry in pairs or ():
                    scorer.add(gt_entry, pred_entry)
                    if save_scores:
                        saved[eval_m].append(pred_entry)
                if 0 <= n_batches <= val_b + 1:
                    break
            table, image_means = scorer.report()
            if want_metrics:
                grand.extend(image_means)
                publish(table)
    if want_metrics:
        publish({'avg/%s_R' % name: np.mean(grand)})
    return saved
