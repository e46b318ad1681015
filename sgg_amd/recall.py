"""Scene-graph recall evaluator on the HIP path -- the interface of lib/sgg_eval.py.

    BasicSceneGraphEvaluator(mode, multiple_preds=False, triplet_counts=None, triplet2str=None, per_triplet=False)
        .evaluate_scene_graph_entry(gt_entry, pred_entry, iou_thresh=0.5)       lib/sgg_eval.py:52-56
        .evaluate_scene_graph_batch(gt_entries, pred_entries, iou_thresh=0.5)   many images, ONE matching launch
        .print_stats() / .result_dict / .save()                                  :57-117
    evaluate_recall(...)                                                         :280-344
    calculate_mR_from_evaluator_list(...), eval_entry(...)                       :420-496

What runs where: the graph-unconstrained ranking (argsort of obj_s * obj_o * rel_scores[:,1:] over every (pair, predicate),
:215-219) is a segmented radix sort on the device (det.hip), the triplet matching (:347-417) is `sgg_recall_first_match`
(eval.hip): one wave per ground-truth triplet scans its image's predictions in rank order.  R@K only depends on the rank
of the FIRST matching prediction of each GT triplet: |union(pred_to_gt[:K])| = #{g : first_rank[g] < K}.
The bookkeeping around it (result_dict layout, printing, per-predicate evaluator lists) is host Python as in the reference.
"""
import pickle

import numpy as np
import torch

from . import _lib, ops

MAX_RECALL_K = 300                                   # lib/sgg_eval.py:12
MODES = ('sgdet', 'sgcls', 'predcls')                # config.py:28
_NO_MATCH = 0x7fffffff
RECALL_KS = (20, 50, 100, 200, 300)
TRIPLET_KS = (5, 10, 15, 20, 50)


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError('sgg_amd.recall needs the GPU (sgg_recall_first_match); there is no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def _i32(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.int32).to(dev).contiguous()


def _f32(a, dev):
    return torch.as_tensor(np.ascontiguousarray(a)).to(torch.float32).to(dev).contiguous()


def intersect_2d(x1, x2):
    """lib/pytorch_misc.py:446-460"""
    if x1.shape[1] != x2.shape[1]:
        raise ValueError('Input arrays must have same #columns')
    return (x1[..., None] == x2.T[None, ...]).all(1)


def argsort_desc(scores):
    """lib/pytorch_misc.py:529-536 on the device: rows of indices into `scores`, best first (ties: lower flat index first,
    where numpy's quicksort leaves the order unspecified)."""
    from .sgdet import _sort_desc
    dev = _dev()
    flat = _f32(np.asarray(scores).ravel(), dev)
    n = flat.numel()
    if n == 0:
        return np.zeros((0, np.asarray(scores).ndim), dtype=np.int64)
    seg = torch.tensor([0, n], dtype=torch.int32, device=dev)
    _, order = _sort_desc(flat, seg, 1)
    return np.column_stack(np.unravel_index(order.cpu().numpy().astype(np.int64), np.asarray(scores).shape))


class _Case(object):
    """One image's matching problem: GT triplets and ranked predicted triplets."""
    __slots__ = ('gt_trip', 'gt_box', 'gt_pair', 'pred_trip', 'pred_box', 'pred_pair')


def _triplet(predicates, relations, classes, boxes):
    """lib/sgg_eval.py:347-378 (without the score columns)."""
    assert predicates.shape[0] == relations.shape[0]
    so = classes[relations[:, :2]]
    trip = np.column_stack((so[:, 0], predicates, so[:, 1])).astype(np.int32)
    tbox = np.column_stack((boxes[relations[:, 0]], boxes[relations[:, 1]])).astype(np.float32)
    return trip, tbox


def _make_case(gt_rels, gt_boxes, gt_classes, pred_rels, pred_boxes, pred_classes):
    c = _Case()
    c.gt_trip, c.gt_box = _triplet(gt_rels[:, 2], gt_rels[:, :2], gt_classes, gt_boxes)
    c.gt_pair = np.ascontiguousarray(gt_rels[:, :2]).astype(np.int32)
    if pred_rels.shape[0]:
        assert pred_rels[:, :2].max() < pred_classes.shape[0]
        assert np.all(pred_rels[:, 2] > 0)
    c.pred_trip, c.pred_box = _triplet(pred_rels[:, 2], pred_rels[:, :2], pred_classes, pred_boxes)
    c.pred_pair = np.ascontiguousarray(pred_rels[:, :2]).astype(np.int32)
    return c


def match_cases(cases, iou_thresh=0.5, phrdet=False, want_pair_rank=False):
    """Runs sgg_recall_first_match over a list of _Case.  -> list of (first_rank i64[G], pair_rank i64[G] or None)."""
    dev = _dev()
    G = sum(c.gt_trip.shape[0] for c in cases)
    if G == 0:
        return [(np.zeros(0, np.int64), np.zeros(0, np.int64) if want_pair_rank else None) for _ in cases]
    cat = lambda xs, shape, dt: np.concatenate(xs, 0) if xs else np.zeros(shape, dt)
    gt_trip = _i32(cat([c.gt_trip for c in cases], (0, 3), np.int32), dev)
    gt_box = _f32(cat([c.gt_box for c in cases], (0, 8), np.float32), dev)
    gt_img = _i32(np.concatenate([np.full(c.gt_trip.shape[0], i, np.int32) for i, c in enumerate(cases)]), dev)
    pred_trip = _i32(cat([c.pred_trip for c in cases], (0, 3), np.int32), dev)
    pred_box = _f32(cat([c.pred_box for c in cases], (0, 8), np.float32), dev)
    ptr = np.zeros(len(cases) + 1, np.int32)
    ptr[1:] = np.cumsum([c.pred_trip.shape[0] for c in cases])
    pred_ptr = _i32(ptr, dev)
    gt_pair = pred_pair = pair_rank = None
    if want_pair_rank:
        gt_pair = _i32(np.concatenate([c.gt_pair for c in cases], 0), dev)
        pred_pair = _i32(cat([c.pred_pair for c in cases], (0, 2), np.int32), dev)
        pair_rank = torch.empty(G, dtype=torch.int32, device=dev)
    first = torch.empty(G, dtype=torch.int32, device=dev)
    p = lambda t: t.data_ptr() if t is not None and t.numel() else None
    _lib.call('sgg_recall_first_match', p(gt_trip), p(gt_box), p(gt_img), G, p(pred_trip), p(pred_box), pred_ptr.data_ptr(),
              len(cases), p(gt_pair), p(pred_pair) if want_pair_rank else None, float(iou_thresh), int(bool(phrdet)),
              first.data_ptr(), p(pair_rank), ops._stream())
    first = first.cpu().numpy().astype(np.int64)
    pr = pair_rank.cpu().numpy().astype(np.int64) if want_pair_rank else None
    out, off = [], 0
    for c in cases:
        g = c.gt_trip.shape[0]
        out.append((first[off:off + g], pr[off:off + g] if want_pair_rank else None))
        off += g
    return out


def _all_matches(case, iou_thresh, phrdet):
    """Complete pred_to_gt lists (lib/sgg_eval.py:380-417) for the single-image API: every match, not only the first.
    Found by re-running the first-match kernel on the suffix after each hit (a GT triplet rarely matches more than a few
    predictions)."""
    P = case.pred_trip.shape[0]
    pred_to_gt = [[] for _ in range(P)]
    start = np.zeros(case.gt_trip.shape[0], np.int64)
    live = np.arange(case.gt_trip.shape[0])
    while live.size:
        subs = []
        for g in live:
            c = _Case()
            c.gt_trip, c.gt_box, c.gt_pair = case.gt_trip[g:g + 1], case.gt_box[g:g + 1], case.gt_pair[g:g + 1]
            s = start[g]
            c.pred_trip, c.pred_box, c.pred_pair = case.pred_trip[s:], case.pred_box[s:], case.pred_pair[s:]
            subs.append(c)
        res = match_cases(subs, iou_thresh, phrdet)
        nxt = []
        for g, (first, _) in zip(live, res):
            if first[0] != _NO_MATCH:
                p = int(start[g] + first[0])
                pred_to_gt[p].append(int(g))
                start[g] = p + 1
                if start[g] < P:
                    nxt.append(g)
        live = np.array(nxt, dtype=np.int64)
    for lst in pred_to_gt:
        lst.sort()
    return pred_to_gt


def evaluate_recall(gt_rels, gt_boxes, gt_classes, pred_rels, pred_boxes, pred_classes, rel_scores=None, cls_scores=None,
                    iou_thresh=0.5, phrdet=False):
    """lib/sgg_eval.py:280-344.  -> (pred_to_gt, pred_5ples, relation_scores)"""
    if pred_rels.size == 0:
        return [[]], np.zeros((0, 5)), np.zeros(0)
    assert gt_rels.shape[0] != 0
    case = _make_case(gt_rels, gt_boxes, gt_classes, pred_rels, pred_boxes, pred_classes)
    pred_to_gt = _all_matches(case, iou_thresh, phrdet)
    relation_scores = None
    if rel_scores is not None and cls_scores is not None:
        relation_scores = np.column_stack((cls_scores[pred_rels[:, 0]], cls_scores[pred_rels[:, 1]], rel_scores))
    pred_5ples = np.column_stack((pred_rels[:, :2], case.pred_trip[:, [0, 2, 1]]))
    return pred_to_gt, pred_5ples, relation_scores


class BasicSceneGraphEvaluator(object):
    def __init__(self, mode, multiple_preds=False, triplet_counts=None, triplet2str=None, per_triplet=False):
        self.mode, self.multiple_preds, self.per_triplet = mode, multiple_preds, per_triplet
        key = lambda name: self.mode + '_' + name
        self.result_dict = {key('recall'): {k: [] for k in RECALL_KS}}
        if per_triplet:                                  # the extra tables of lib/sgg_eval.py:25-40
            rd = self.result_dict
            rd[key('recall_norm')] = {k: [] for k in RECALL_KS}
            rd[key('rank')], rd[key('counts')] = [], []
            for sfx in ('', '_norm'):
                rd[key('recall_triplet' + sfx)] = {k: [] for k in TRIPLET_KS}
                rd[key('meanrank_triplet' + sfx)] = []
                rd[key('medianrank_triplet' + sfx)] = []
            rd[key('medianrankclass_triplet')] = []
            self.triplet_counts, self.triplet2str, self.triplet_ranks = triplet_counts, triplet2str, {}

    @classmethod
    def all_modes(cls, **kwargs):
        return {m: cls(mode=m, **kwargs) for m in MODES}

    @classmethod
    def vrd_modes(cls, **kwargs):
        return {m: cls(mode=m, multiple_preds=True, **kwargs) for m in ('preddet', 'phrdet')}

    def save(self, fn):
        np.save(fn, self.result_dict)

    def normalize_counts(self, counts):
        weights = 1. / (counts + 1)
        weights /= weights.sum()
        return weights

    # ------------------------------------------------------------------ per-image preparation (:134-222)
    def _prepare(self, gt_entry, pred_entry, mode, multiple_preds, ranked=None):
        """-> None (entry fully handled on the host: preddet) or a dict for the matching stage.  `ranked` (optional dict): the
        prediction ranking depends on pred_entry only, not on the ground truth -- evaluators that see the same image (the 2 x 50
        per-predicate evaluators) share it through this cache instead of sorting the image's scores again."""
        result_dict = self.result_dict
        gt_rels = np.asarray(gt_entry['gt_relations'])
        gt_boxes = np.asarray(gt_entry['gt_boxes']).astype(float)
        gt_classes = np.asarray(gt_entry['gt_classes'])
        pred_rel_inds = np.asarray(pred_entry['pred_rel_inds'])
        rel_scores = np.asarray(pred_entry['rel_scores'])
        if mode == 'predcls':
            pred_boxes, pred_classes, obj_scores = gt_boxes, gt_classes, np.ones(gt_classes.shape[0])
        elif mode == 'sgcls':
            pred_boxes, pred_classes, obj_scores = gt_boxes, np.asarray(pred_entry['pred_classes']), np.asarray(pred_entry['obj_scores'])
        elif mode == 'objcls':
            pred_boxes, pred_classes, obj_scores = gt_boxes, np.asarray(pred_entry['pred_classes']), np.asarray(pred_entry['obj_scores'])
            pred_rel_inds = gt_rels[:, :2]                       # perfect predicate recognition (:158-161)
            rel_scores = np.zeros((len(gt_rels), rel_scores.shape[1]))
            rel_scores[np.arange(len(gt_rels)), gt_rels[:, 2]] = 1
        elif mode == 'sgdet' or mode == 'phrdet':
            pred_boxes = np.asarray(pred_entry['pred_boxes']).astype(float)
            pred_classes, obj_scores = np.asarray(pred_entry['pred_classes']), np.asarray(pred_entry['obj_scores'])
        elif mode == 'preddet':                                  # :167-194, host only
            # predicate detection: the predictions on the GROUND-TRUTH pairs only -- for every gt pair the first prediction row on that pair
            pair_hits = intersect_2d(pred_rel_inds, gt_rels[:, :2])                     # [predictions, gt pairs]
            recalls = (result_dict[mode + '_recall'], result_dict[mode + '_recall_norm'] if self.per_triplet else {})
            if pair_hits.size == 0:
                for table in recalls:
                    for k in table:
                        table[k].append(0.0)
                return None
            rows = np.argmax(pair_hits, axis=0)                                         # (first matching prediction per gt pair)
            on_gt_pairs, their_scores = pred_rel_inds[rows], rel_scores[rows]
            by_score = argsort_desc(their_scores[:, 1:])                                # (pair row, predicate - 1), best first
            ranked_rels = np.concatenate((on_gt_pairs[by_score[:, 0]], by_score[:, 1:2] + 1), axis=1)    # (not `ranked`: that is the cache parameter)
            found = intersect_2d(ranked_rels, gt_rels)
            n_gt = float(gt_rels.shape[0])
            for table in recalls:
                for k in table:
                    table[k].append(float(found[:k].any(0).sum()) / n_gt)
            return None
        else:
            raise ValueError('invalid mode')

        # the ranking cache is keyed on the prediction entry; in objcls mode the ranked relations are derived from the GROUND TRUTH
        # (and callers build temporary gt dicts whose ids CPython may reuse within one cache's lifetime): never cached there
        if mode == 'objcls':
            ranked = None
        key = (id(pred_entry), mode, bool(multiple_preds))
        hit = ranked.get(key) if ranked is not None else None
        if hit is not None:
            pred_rels, predicate_scores, overall_order = hit
        elif multiple_preds:                                     # :215-219
            obj_scores_per_rel = obj_scores[pred_rel_inds].prod(1)
            overall_scores = obj_scores_per_rel[:, None] * rel_scores[:, 1:]
            overall_order = argsort_desc(overall_scores)
            score_inds = overall_order[:MAX_RECALL_K]
            pred_rels = np.column_stack((pred_rel_inds[score_inds[:, 0]], score_inds[:, 1] + 1))
            predicate_scores = rel_scores[score_inds[:, 0], score_inds[:, 1] + 1]
        else:                                                    # :220-222
            overall_order = None
            fg = rel_scores[:, 1:]                                                     # graph constraint: one predicate per pair, the best non-background one
            best = fg.argmax(1)
            pred_rels = np.concatenate((pred_rel_inds, best[:, None] + 1), axis=1)
            predicate_scores = fg[np.arange(len(fg)), best]
        if ranked is not None and hit is None:
            ranked[key] = (pred_rels, predicate_scores, overall_order)
        prep = dict(gt_rels=gt_rels, gt_boxes=gt_boxes, gt_classes=gt_classes, pred_rels=pred_rels, pred_boxes=pred_boxes,
                    pred_classes=pred_classes, predicate_scores=predicate_scores, obj_scores=obj_scores, mode=mode,
                    pred_rel_inds=pred_rel_inds, overall_order=overall_order)
        return prep

    def _account(self, prep, first_rank, pair_rank):
        """Recall bookkeeping of one image (:233-277) from the first-match ranks."""
        mode, result_dict = prep['mode'], self.result_dict
        gt_rels, gt_classes = prep['gt_rels'], prep['gt_classes']
        G = gt_rels.shape[0]
        weights = counts = None
        if self.per_triplet:
            counts = np.zeros(G)
            for i, (o, s, R) in enumerate(gt_rels):
                tri = '{}_{}_{}'.format(gt_classes[o], R, gt_classes[s])
                if tri in self.triplet_counts:
                    counts[i] = self.triplet_counts[tri]
            weights = self.normalize_counts(counts)
        for k in result_dict[mode + '_recall']:
            hit = first_rank < k
            result_dict[mode + '_recall'][k].append(float(hit.sum()) / float(G))
            if self.per_triplet:
                result_dict[mode + '_recall_norm'][k].append(np.sum(weights[hit]))
        if self.per_triplet:
            ranks = np.where(pair_rank >= 0, pair_rank, MAX_RECALL_K + 1).astype(np.float64)
            for i, (o, s, R) in enumerate(gt_rels):
                tri = '{}_{}_{}'.format(gt_classes[o], R, gt_classes[s])
                self.triplet_ranks.setdefault(tri, []).append(ranks[i])
            result_dict[mode + '_rank'].extend(ranks)
            result_dict[mode + '_counts'].extend(counts)

    # ------------------------------------------------------------------ public evaluation entry points
    def evaluate_scene_graph_batch(self, gt_entries, pred_entries, iou_thresh=0.5, ranked=None):
        """Many images, one matching launch (and one more for the per-triplet ranks).  Updates result_dict exactly as
        calling evaluate_scene_graph_entry image by image.  `ranked`: see _prepare."""
        if self.per_triplet and not self.multiple_preds:
            raise NameError('per_triplet needs multiple_preds=True (overall_scores, lib/sgg_eval.py:249)')
        preps = [self._prepare(g, p, self.mode, self.multiple_preds, ranked) for g, p in zip(gt_entries, pred_entries)]
        live = [p for p in preps if p is not None]
        if not live:
            return
        cases = [_make_case(p['gt_rels'], p['gt_boxes'], p['gt_classes'], p['pred_rels'], p['pred_boxes'], p['pred_classes'])
                 for p in live]
        res = match_cases(cases, iou_thresh, phrdet=(self.mode == 'phrdet'))
        pair = [None] * len(live)
        if self.per_triplet:                                     # ranks among ALL (pair, predicate) scores (:249-272)
            full = []
            for p in live:
                o = p['overall_order']
                rels_all = np.column_stack((p['pred_rel_inds'][o[:, 0]], o[:, 1] + 1))
                full.append(_make_case(p['gt_rels'], p['gt_boxes'], p['gt_classes'], rels_all, p['pred_boxes'], p['pred_classes']))
            pair = [r[1] for r in match_cases(full, 0.5, phrdet=False, want_pair_rank=True)]
        for p, (first, _), pr in zip(live, res, pair):
            self._account(p, first, pr)

    def evaluate_scene_graph_entry(self, gt_entry, pred_scores, viz_dict=None, iou_thresh=0.5):
        return self.evaluate_from_dict(gt_entry, pred_scores, self.mode, self.result_dict, viz_dict=viz_dict,
                                       iou_thresh=iou_thresh, multiple_preds=self.multiple_preds)

    def evaluate_from_dict(self, gt_entry, pred_entry, mode, result_dict, multiple_preds=False, viz_dict=None, **kwargs):
        """lib/sgg_eval.py:120-277 for one image; returns (pred_to_gt, pred_5ples, rel_scores) like the reference."""
        assert result_dict is self.result_dict
        if self.per_triplet and not multiple_preds:
            raise NameError('per_triplet needs multiple_preds=True (overall_scores, lib/sgg_eval.py:249)')
        prep = self._prepare(gt_entry, pred_entry, mode, multiple_preds)
        if prep is None:
            return None, None, None
        iou_thresh = kwargs.get('iou_thresh', 0.5)
        pred_to_gt, pred_5ples, rel_scores = evaluate_recall(
            prep['gt_rels'], prep['gt_boxes'], prep['gt_classes'], prep['pred_rels'], prep['pred_boxes'], prep['pred_classes'],
            prep['predicate_scores'], prep['obj_scores'], phrdet=(mode == 'phrdet'), iou_thresh=iou_thresh)
        G = prep['gt_rels'].shape[0]
        first = np.full(G, _NO_MATCH, np.int64)
        for p, lst in enumerate(pred_to_gt):
            for g in lst:
                first[g] = min(first[g], p)
        pr = None
        if self.per_triplet:
            o = prep['overall_order']
            rels_all = np.column_stack((prep['pred_rel_inds'][o[:, 0]], o[:, 1] + 1))
            full = _make_case(prep['gt_rels'], prep['gt_boxes'], prep['gt_classes'], rels_all, prep['pred_boxes'], prep['pred_classes'])
            pr = match_cases([full], 0.5, phrdet=False, want_pair_rank=True)[0][1]
        self._account(prep, first, pr)
        return pred_to_gt, pred_5ples, rel_scores

    # ------------------------------------------------------------------ reporting (:65-117)
    def _tri(self, o, s, R, gt_classes):
        return '{}_{}_{}'.format(gt_classes[o], R, gt_classes[s])

    def print_stats(self, verbose=True):
        """Image-level R@K means (returned as {'R@K': value} when verbose, as the reference does) and, with per_triplet,
        the triplet-level tables written back into result_dict."""
        rd, key = self.result_dict, (lambda name: self.mode + '_' + name)
        recalls = rd[key('recall')]
        output = {}
        if verbose:
            n_img = len(next(iter(recalls.values())))
            print('================%s%s: %d images ==================' % (self.mode, '(NO GC)' if self.multiple_preds else '(GC)', n_img))
            for k, v in recalls.items():
                output['R@%i' % k] = np.mean(v)
                print('R@%i: %f' % (k, output['R@%i' % k]))
        if not self.per_triplet:
            return output
        ranks = np.asarray(rd[key('rank')], dtype=np.float32)
        w = self.normalize_counts(np.asarray(rd[key('counts')], dtype=np.float32))
        if verbose:
            print('\nTriplet level evaluation (%d triplets)' % len(ranks))
        for k in TRIPLET_KS:
            hit = ranks < k
            rd[key('recall_triplet')][k] = hit.mean()
            rd[key('recall_triplet_norm')][k] = (hit.astype(np.float32) * w).sum()
            if verbose:
                print('Triplet level R@%i: %.4f (normalized: %.4f)' % (k, rd[key('recall_triplet')][k], rd[key('recall_triplet_norm')][k]))
        # median rank per triplet class, weighted by inverse training frequency
        cls_medians = np.array([np.median(v) for v in self.triplet_ranks.values() if len(v)])
        cls_counts = np.array([self.triplet_counts.get(t, 0) for t, v in self.triplet_ranks.items() if len(v)])
        cw = self.normalize_counts(cls_counts)
        rd[key('meanrank_triplet')] = ranks.mean()
        rd[key('meanrank_triplet_norm')] = (ranks * w).sum()
        rd[key('medianrank_triplet')] = np.median(ranks)
        rd[key('medianrankclass_triplet')] = cls_medians.mean()
        rd[key('medianrank_triplet_norm')] = (cls_medians * cw).sum()
        if verbose:
            print('Triplet level mean rank: %.4f (normalized: %.4f)' % (rd[key('meanrank_triplet')], rd[key('meanrank_triplet_norm')]))
            print('Triplet level median rank: %.4f (per class: %.4f, normalized per class: %.4f)\n'
                  % (rd[key('medianrank_triplet')], rd[key('medianrankclass_triplet')], rd[key('medianrank_triplet_norm')]))
        return output


class PredicateRecall(object):
    """Recall per predicate and its mean over predicates -- the job of lib/sgg_eval.py:420-496 (calculate_mR_from_evaluator_list,
    eval_entry).  One graph-constrained and one unconstrained evaluator set per foreground predicate; a predicate's evaluators see
    an image only if its ground truth holds that predicate, and then only those ground-truth relations.  Images arrive in
    batches: per predicate ONE matching launch per evaluator over the images that contain it, and the prediction ranking of an
    image is computed once and shared by all predicates."""

    def __init__(self, ind_to_predicates=None):
        self.rows = []                                   # (predicate id, name, {mode: evaluator}, {mode: unconstrained evaluator})
        for pid, pname in enumerate(ind_to_predicates or ()):
            if pid > 0:                                  # 0 = background
                self.rows.append((pid, pname, BasicSceneGraphEvaluator.all_modes(),
                                  BasicSceneGraphEvaluator.all_modes(multiple_preds=True)))

    @classmethod
    def from_lists(cls, evaluator_list, evaluator_multiple_preds_list):
        """Wraps the reference's two parallel lists [(id, name, {mode: evaluator})]."""
        self = cls()
        for (pid, pname, ev), (pid2, _, ev_mp) in zip(evaluator_list, evaluator_multiple_preds_list):
            assert pid == pid2
            self.rows.append((pid, pname, ev, ev_mp))
        return self

    def evaluate_batch(self, mode, gt_entries, pred_entries):
        holders = {}                                     # predicate id -> images whose ground truth holds it
        for i, g in enumerate(gt_entries):
            for pid in np.unique(np.asarray(g['gt_relations'])[:, -1]):
                holders.setdefault(int(pid), []).append(i)
        ranked = {}
        for pid, _, ev, ev_mp in self.rows:
            imgs = holders.get(pid)
            if not imgs:
                continue
            sub_gt = []
            for i in imgs:
                rel = np.asarray(gt_entries[i]['gt_relations'])
                sub_gt.append(dict(gt_entries[i], gt_relations=rel[rel[:, -1] == pid]))
            sub_pred = [pred_entries[i] for i in imgs]
            ev[mode].evaluate_scene_graph_batch(sub_gt, sub_pred, ranked=ranked)
            ev_mp[mode].evaluate_scene_graph_batch(sub_gt, sub_pred, ranked=ranked)

    def table(self, mode, multiple_preds=False):
        """-> (names, f64[n_predicates, len(RECALL_KS)]): mean image-level R@K of every predicate (NaN: predicate never seen)."""
        names, rows = [], []
        for _, pname, ev, ev_mp in self.rows:
            rec = (ev_mp if multiple_preds else ev)[mode].result_dict[mode + '_recall']
            names.append(pname)
            rows.append([np.mean(rec[k]) if len(rec[k]) else np.nan for k in RECALL_KS])
        return names, np.asarray(rows, dtype=np.float64).reshape(len(names), len(RECALL_KS))

    def mean_recall(self, mode, multiple_preds=False, save_file=None, verbose=True):
        """{'R@K': mR@K}.  As in the reference, a predicate without any image (NaN R@100) adds nothing to the sum but still
        counts in the denominator."""
        names, tab = self.table(mode, multiple_preds)
        seen = ~np.isnan(tab[:, RECALL_KS.index(100)]) if len(names) else np.zeros(0, bool)
        mr = np.where(seen[:, None], tab, 0.0).sum(0) / max(len(names), 1)
        mean_recall = {'R@%d' % k: float(v) for k, v in zip(RECALL_KS, mr)}
        if verbose:
            print('\n====== %s  mean recall %s constraint: %d predicates, %d seen ======' %
                  (mode, 'without' if multiple_preds else 'with', len(names), int(seen.sum())))
            for pname, row in zip(names, tab):
                print('  %-16s ' % pname + '  '.join('R@%d %.4f' % (k, v) for k, v in zip(RECALL_KS, row)))
            print('  ' + '  '.join('mR@%d: %.6f' % (k, mean_recall['R@%d' % k]) for k in RECALL_KS[:4]))
        if save_file is not None:
            if multiple_preds:
                save_file = save_file.replace('.pkl', '_multiple_preds.pkl')
            per_pred = {pname: {'R@%d' % k: float(v) for k, v in zip(RECALL_KS, row)} for pname, row in zip(names, tab)}
            per_pred['mean_recall'] = mean_recall
            with open(save_file, 'wb') as f:
                pickle.dump(per_pred, f)
        return mean_recall


def calculate_mR_from_evaluator_list(evaluator_list, mode, multiple_preds=False, save_file=None):
    """The reference's entry point (lib/sgg_eval.py:420-478) over ONE list [(id, name, {mode: evaluator})]; `multiple_preds` only
    says which kind of evaluators the list holds (it names the table and the save file)."""
    pr = PredicateRecall.from_lists(evaluator_list, evaluator_list)
    return pr.mean_recall(mode, multiple_preds=multiple_preds, save_file=save_file)


def eval_entry(mode, gt_entry, pred_entry, evaluator_list, evaluator_multiple_preds_list):
    """The reference's per-image entry point (lib/sgg_eval.py:481-496)."""
    PredicateRecall.from_lists(evaluator_list, evaluator_multiple_preds_list).evaluate_batch(mode, [gt_entry], [pred_entry])
