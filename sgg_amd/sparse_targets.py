"""FrequencyBias: P(predicate | class_subj, class_obj) from training-set counts, the `-use_bias` / `-test_bias` flags.

Mirrors lib/sparse_targets.py (FrequencyBias) and lib/get_dataset_counts.py (get_counts, box_filter).  The counts are a one-off
host pass over the training split at construction (numpy, as in the reference); the per-step lookup
`index_with_labels` and its gradient into the embedding are HIP kernels (csrc/freq.hip)."""
import numpy as np
import torch
import torch.nn as nn

from . import _lib, ops


def _iou_positive(boxes):
    """[N,N] bool: IoU > 0 (lib/pytorch_misc.py:60-67 bbox_overlaps -> [3P] box_iou, only its sign is used)."""
    b = np.asarray(boxes, dtype=np.float32)
    lt = np.maximum(b[:, None, :2], b[None, :, :2])
    rb = np.minimum(b[:, None, 2:], b[None, :, 2:])
    wh = np.clip(rb - lt, 0, None)
    inter = wh[..., 0] * wh[..., 1]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    with np.errstate(divide='ignore', invalid='ignore'):
        return inter / (area[:, None] + area[None] - inter) > 0


def box_filter(boxes, must_overlap=False):
    """lib/get_dataset_counts.py:46-64"""
    n = boxes.shape[0]
    every = ~np.eye(n, dtype=bool)
    if must_overlap:
        ov = _iou_positive(boxes)
        np.fill_diagonal(ov, False)
        cand = np.column_stack(np.where(ov))
        if cand.size == 0:
            cand = np.column_stack(np.where(every))
        return cand
    return np.column_stack(np.where(every))


def get_counts(train_data, must_overlap=True):
    """lib/get_dataset_counts.py:10-43 -> fg_matrix i64[C,C,P], bg_matrix i64[C,C].  train_data: .num_classes,
    .num_predicates, per-image lists .gt_classes, .relationships ([R,3] = subj, obj, predicate), .gt_boxes."""
    C, P = train_data.num_classes, train_data.num_predicates
    fg = np.zeros((C, C, P), dtype=np.int64)
    bg = np.zeros((C, C), dtype=np.int64)
    for i in range(len(train_data)):
        cls = np.asarray(train_data.gt_classes[i])
        rels = np.asarray(train_data.relationships[i])
        o1o2 = cls[rels[:, :2]]
        np.add.at(fg, (o1o2[:, 0], o1o2[:, 1], rels[:, 2]), 1)
        cand = cls[np.array(box_filter(np.asarray(train_data.gt_boxes[i]), must_overlap=must_overlap), dtype=int)]
        np.add.at(bg, (cand[:, 0], cand[:, 1]), 1)
    return fg, bg


class _IndexWithLabels(torch.autograd.Function):
    """table rows gathered per edge (+ rel_dists); backward scatters into the dense embedding gradient."""

    @staticmethod
    def forward(ctx, table, rel_dists, obj_dists, gt_classes, rel_inds, num_objs):
        E, P = rel_inds.shape[0], table.shape[1]
        N = obj_dists.shape[0] if obj_dists is not None else gt_classes.shape[0]
        dev = table.device
        preds = torch.empty(N, dtype=torch.int64, device=dev)
        row_idx = torch.empty(E, dtype=torch.int32, device=dev)
        out = torch.empty(E, P, dtype=torch.float32, device=dev)
        od = obj_dists.contiguous() if obj_dists is not None else None
        ri = rel_dists.float().contiguous() if rel_dists is not None else None
        _lib.call('sgg_freq_bias_fwd', ops._p(od) if od is not None else None, N, num_objs,
                  ops._p(gt_classes.contiguous(), torch.int64) if gt_classes is not None else None,
                  ops._p(rel_inds.contiguous(), torch.int64), E, ops._p(table.contiguous(), torch.float32), P,
                  ops._p(ri) if ri is not None else None, ops._p(out), ops._p(preds), ops._p(row_idx),
                  ops.dt(od) if od is not None else _lib.SGG_F32, ops._stream())
        ctx.save_for_backward(row_idx)
        ctx.shape = tuple(table.shape)
        ctx.has_rel = rel_dists is not None
        ctx.mark_non_differentiable(preds)
        return out, preds

    @staticmethod
    def backward(ctx, d_out, _d_preds):
        row_idx, = ctx.saved_tensors
        d_table = None
        if ctx.needs_input_grad[0]:
            d_table = torch.zeros(ctx.shape, dtype=torch.float32, device=d_out.device)
            g = d_out.float().contiguous()
            _lib.call('sgg_freq_bias_bwd', ops._p(g), ops._p(row_idx), g.shape[0], g.shape[1], ops._p(d_table), ops._stream())
        return d_table, (d_out if ctx.has_rel else None), None, None, None, None


class FrequencyBias(nn.Module):
    """lib/sparse_targets.py:7-48.  `obj_baseline.weight` f32[C*C, P] is a trainable embedding, as in the reference."""

    def __init__(self, train_data, eps=1e-3):
        super(FrequencyBias, self).__init__()
        fg_matrix, bg_matrix = get_counts(train_data, must_overlap=True)
        bg_matrix += 1
        fg_matrix[:, :, 0] = bg_matrix
        pred_dist = np.log(fg_matrix / fg_matrix.sum(2)[:, :, None] + eps)
        self.num_objs = pred_dist.shape[0]
        pred_dist = torch.as_tensor(pred_dist, dtype=torch.float32).view(-1, pred_dist.shape[2])
        self.obj_baseline = nn.Embedding(pred_dist.size(0), pred_dist.size(1))
        self.obj_baseline.weight.data = pred_dist

    def index_with_labels(self, labels):
        """labels i64[E,2] = (class_subj, class_obj) -> f32[E,P]  (lib/sparse_targets.py:26-31)"""
        E = labels.shape[0]
        flat = labels.reshape(-1).contiguous()                       # boxes 2e, 2e+1 carry edge e's two classes
        rel = torch.arange(2 * E, device=labels.device, dtype=torch.int64).view(E, 2)
        rel = torch.cat((torch.zeros(E, 1, dtype=torch.int64, device=labels.device), rel), 1)
        return _IndexWithLabels.apply(self.obj_baseline.weight, None, None, flat, rel, self.num_objs)[0]

    def apply_to(self, rel_dists, obj_dists, rel_inds, gt_classes=None, replace=False):
        """The use_bias block of RelModelStanford.forward (rel_model_stanford.py:159-177) in one call:
        obj_preds from softmax(obj_dists) (or gt_classes i64[N] for predcls), rel_dists + bias (or the bias alone, test_bias).
        -> (rel_dists f32[E,P], obj_preds i64[N])"""
        return _IndexWithLabels.apply(self.obj_baseline.weight, None if replace else rel_dists,
                                      obj_dists if gt_classes is None else None, gt_classes, rel_inds, self.num_objs)

    def forward(self, obj_cands0, obj_cands1):
        raise NotImplementedError('FrequencyBias.forward (joint-distribution form, lib/sparse_targets.py:33-48) is not called by '
                                  'RelModelStanford; only index_with_labels is on the path')
