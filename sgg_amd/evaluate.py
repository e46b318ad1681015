"""Evaluation harness around the path: lib/eval.py val_epoch / val_batch and lib/pytorch_misc.py set_mode.

Same call signatures and result keys as the reference; the forward is the HIP path (`sgg_model(b.scatter())`, one image per
call as in dataloaders/visual_genome.py:730) and the recall accounting is `sgg_amd.recall` (HIP matching kernel).  Not carried
over: the matplotlib / networkx visualisation branch (`vis=True`, lib/eval.py:180-221) and wandb itself -- `wandb_log` may be
any callable taking (dict, step=..., is_summary=..., log_repeats=...)."""
import numpy as np
import torch

from .recall import BasicSceneGraphEvaluator, calculate_mR_from_evaluator_list, eval_entry
from .sparse_targets import get_counts

IM_SCALE = 592      # config.py:31
BOX_SCALE = 1024    # config.py:30
all_shot_splits = ['val_alls', 'test_alls']          # lib/eval.py:12


def set_mode(sgg_model, mode, is_train, verbose=False):
    """lib/pytorch_misc.py:76-95"""
    if is_train:
        sgg_model.train()
    else:
        sgg_model.eval()
    sgg_model.mode = mode
    if hasattr(sgg_model, 'detector'):
        m = 'refinerels' if mode == 'sgdet' else 'gtbox'
        if verbose:
            print('setting %s mode for detector' % m)
        sgg_model.detector.mode = m
    if hasattr(sgg_model, 'context'):
        if verbose:
            print('setting %s mode for context' % mode)
        sgg_model.context.mode = mode


def predicate_weights_from(train, predicate_weight):
    """lib/eval.py:25-30: mean predicate frequency ** predicate_weight (background slot = bg count + 1, everything + 1)"""
    fg_matrix, bg_matrix = get_counts(train, must_overlap=True)
    fg_matrix[:, :, 0] = bg_matrix + 1
    fg_matrix = fg_matrix + 1
    return fg_matrix.mean(axis=(0, 1)) ** predicate_weight


def val_batch(sgg_model, batch_num, b, evaluator, eval_m, val_dataset, evaluator_list, evaluator_multiple_preds_list,
              vis=False, max_obj=10, max_rels=20, train=None, test_zs=None, predicate_weights=None):
    """lib/eval.py:120-227.  -> list of pred_entry dicts (one per image of the batch), or None if no threshold gave boxes."""
    if vis:
        raise NotImplementedError('vis=True (matplotlib / networkx drawing, lib/eval.py:180-221) is outside the path')
    if val_dataset.torch_detector:
        scale = 1.
        box_threshs = [0.2, 0.05, 0.01]
    else:
        scale = BOX_SCALE / IM_SCALE
        box_threshs = [None]
    pred_entries = []
    for box_score_thresh in box_threshs:
        sgg_model.set_box_score_thresh(box_score_thresh)
        try:
            det_res = [sgg_model(b.scatter())]
            for i, (boxes_i, objs_i, obj_scores_i, rels_i, pred_scores_i) in enumerate(det_res):
                if getattr(val_dataset, 'split', 'stanford') == 'stanford':                 # VG.split, lib/eval.py:143-147
                    w, h = b[i][1][0, :2]
                    scale_gt = 1. / (BOX_SCALE / max(w, h))
                else:
                    scale_gt = 1.
                gt_entry = {
                    'gt_classes': val_dataset.gt_classes[batch_num + i].copy(),
                    'gt_relations': val_dataset.relationships[batch_num + i].copy(),
                    'gt_boxes': val_dataset.gt_boxes[batch_num + i].copy() * scale_gt,
                }
                pred_entry = {
                    'pred_boxes': boxes_i * scale,
                    'pred_classes': objs_i,
                    'pred_rel_inds': rels_i,
                    'obj_scores': obj_scores_i,
                    'rel_scores': pred_scores_i,
                }
                if predicate_weights is not None:                                            # :163-167
                    p = 1. / predicate_weights[1:]
                    pred_entry['rel_scores'][:, 1:] = pred_entry['rel_scores'][:, 1:] * p
                    pred_entry['rel_scores'] = pred_entry['rel_scores'] / np.sum(pred_entry['rel_scores'], axis=1,
                                                                                 keepdims=True)
                    assert (abs(pred_entry['rel_scores'].sum(1) - 1) < 1e-5).all(), pred_entry['rel_scores'].sum(1)
                pred_entries.append(pred_entry)
                for sfx in ['', '_nogc']:
                    evaluator[eval_m + sfx].evaluate_scene_graph_entry(gt_entry, pred_entry)
                if evaluator_list is not None and len(evaluator_list) > 0:
                    eval_entry(eval_m, gt_entry, pred_entry, evaluator_list, evaluator_multiple_preds_list)
            return pred_entries
        except (ValueError, IndexError) as e:                                                # :223-227
            print('no objects or relations found'.upper(), e, b[0][-1], 'trying a smaller threshold')
    return None


def val_epoch(mode, sgg_model, loader, name, triplet_counts, triplet2str, n_batches=-1, is_test=False, save_scores=False,
              predicate_weight=0, train=None, wandb_log=None, results=None, **kwargs):
    """lib/eval.py:15-117.  `results` (optional dict, not in the reference) receives the keys the reference hands to wandb
    ('sgcls/test_R@50_GC', 'sgcls/test_mR@50_NOGC', 'avg/test_R', ...) whether or not `wandb_log` is given."""
    print('\nEvaluate %s %s triplets' % (name.upper(), 'test' if is_test else 'val'))
    sgg_model.eval()
    evaluator, all_pred_entries, all_metrics = {}, {}, []
    EVAL_MODES = ['sgdet'] if mode == 'sgdet' else ['predcls', 'sgcls']
    assert mode in EVAL_MODES, (mode, 'other modes not supported')
    predicate_weights = predicate_weights_from(train, predicate_weight) if predicate_weight != 0 else None
    step = getattr(sgg_model, 'global_batch_iter', 0)

    def log(d):
        if results is not None:
            results.update(d)
        if wandb_log:
            wandb_log(d, step=step, is_summary=True, log_repeats=5 if is_test else 1)

    with torch.no_grad():
        for eval_m in EVAL_MODES:
            if eval_m == 'sgdet' and name.find('val_') >= 0:
                continue
            print('\nEvaluating %s...' % eval_m.upper())
            evaluator[eval_m] = BasicSceneGraphEvaluator(eval_m)
            evaluator[eval_m + '_nogc'] = BasicSceneGraphEvaluator(eval_m, multiple_preds=True,
                                                                   per_triplet=name in all_shot_splits,
                                                                   triplet_counts=triplet_counts, triplet2str=triplet2str)
            evaluator_list, evaluator_multiple_preds_list = [], []
            if name not in ['val_zs', 'test_zs'] and name.find('val_') < 0:
                for index, name_s in enumerate(loader.dataset.ind_to_predicates):
                    if index == 0:
                        continue
                    evaluator_list.append((index, name_s, BasicSceneGraphEvaluator.all_modes()))
                    evaluator_multiple_preds_list.append((index, name_s, BasicSceneGraphEvaluator.all_modes(multiple_preds=True)))
            set_mode(sgg_model, mode=eval_m, is_train=False, verbose=True)
            all_pred_entries[eval_m] = []
            for val_b, batch in enumerate(loader):
                pred_entry = val_batch(sgg_model, val_b, batch, evaluator, eval_m, loader.dataset, evaluator_list,
                                       evaluator_multiple_preds_list, train=train, predicate_weights=predicate_weights, **kwargs)
                if save_scores:
                    all_pred_entries[eval_m].extend(pred_entry)
                if n_batches > -1 and val_b + 1 >= n_batches:
                    break
            evaluator[eval_m].print_stats()
            evaluator[eval_m + '_nogc'].print_stats()
            mean_recall = mean_recall_mp = None
            if len(evaluator_list) > 0:
                mean_recall = calculate_mR_from_evaluator_list(evaluator_list, eval_m, save_file=None)
                mean_recall_mp = calculate_mR_from_evaluator_list(evaluator_multiple_preds_list, eval_m, multiple_preds=True,
                                                                  save_file=None)
            if not wandb_log and results is None:
                continue
            eval_gc = evaluator[eval_m].result_dict
            eval_no_gc = evaluator[eval_m + '_nogc'].result_dict
            results_dict = {}
            for eval_, mean_eval, sfx in zip([eval_gc, eval_no_gc], [mean_recall, mean_recall_mp], ['GC', 'NOGC']):
                for k, v in eval_[eval_m + '_recall'].items():
                    all_metrics.append(np.mean(v))
                    results_dict['%s/%s_R@%i_%s' % (eval_m, name, k, sfx)] = np.mean(v)
                if mean_eval:
                    for k, v in mean_eval.items():
                        results_dict['%s/%s_m%s_%s' % (eval_m, name, k, sfx)] = np.mean(v)
            try:                                                                             # per-triplet metrics, :96-107
                if name in all_shot_splits:
                    for case in ['', '_norm']:
                        for k, v in eval_no_gc[eval_m + '_recall_triplet' + case].items():
                            results_dict['%s/%s_R@%i_triplet%s' % (eval_m, name, k, case)] = v
                        for metric in ['meanrank', 'medianrank'] + (['medianrankclass'] if case == '' else []):
                            results_dict['%s/%s_%s_triplet%s' % (eval_m, name, metric, case)] = \
                                eval_no_gc[eval_m + ('_%s_triplet' % metric) + case]
            except Exception as e:
                print('error in per triplet eval', e)
            log(results_dict)
    if wandb_log or results is not None:
        log({'avg/%s_R' % (name): np.mean(all_metrics)})
    return all_pred_entries
