"""Synthetic VG-shaped inputs and random-init weights (there are no datasets / checkpoints offline).
Definition of the synthetic workload: SURVEY.md 8(d) / BASELINE.md section 4."""
import numpy as np
import torch
import torch.nn as nn


class SyntheticData(object):
    """Stand-in for the `train_data` argument: only `.ind_to_classes` / `.ind_to_predicates` are read
    (sgg_models/rel_model_base.py:45-46)."""
    ind_to_classes = ['__background__'] + ['class%d' % i for i in range(1, 151)]
    ind_to_predicates = ['__background__'] + ['pred%d' % i for i in range(1, 51)]


def init_weights(model, seed=111):
    """He-normal init for conv / linear weights so that activations stay O(1) through VGG-16 and the heads
    (PyTorch's default init shrinks them ~1000x over 13 conv layers, which would make parity checks vacuous)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, m in model.named_modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                fan_in = m.weight[0].numel()
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5)
                if m.bias is not None:
                    m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.05)
            elif isinstance(m, nn.GRUCell):
                for p in m.parameters():
                    p.copy_(torch.randn(p.shape, generator=g) * (1.0 / m.hidden_size) ** 0.5)
            elif isinstance(m, nn.BatchNorm2d):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    return model


def spread_detector_(sd, grow=6.0, gain=2.0, seed=17):
    """SGDet (BASELINE configs[2]) needs a detector that BEHAVES like a trained one: ~1000 proposals per image after the RPN's NMS and up
    to 50 detections with distinct scores.  A He-initialised RPN does not: its box deltas are O(1) per anchor side (every proposal is
    clipped to the frame) and its objectness is dominated by a per-anchor constant, so the 1000 best anchors are 1000 shifted copies
    of one large anchor and NMS 0.7 leaves ~13 of them.  In place, on a state dict (or anything with those keys):
      * rpn.head.bbox_pred x 0.02 (proposals stay near their anchors), rpn.head.cls_logits.weight x 0.25 and a bias that puts the 32-px
        anchors (3 ratios x 1444 positions, mutual IoU <= 0.56) ahead of the others: the 1000 best survive NMS 0.7 -> 1000 proposals/img;
      * roi_heads.box_predictor: cls_score x `gain` with a random bias (peaked but distinct softmax scores), bbox_pred x 0.2 with a
        log(`grow`) bias on (dw, dh) so that detections grow to object size and overlap (the overlap filter then keeps ~500 edges/img).
    Same arithmetic, same kernels; only the synthetic weights change.  Returns sd."""
    import math
    pre = 'detector.rpn.head.'
    sd[pre + 'bbox_pred.weight'].mul_(0.02)
    sd[pre + 'bbox_pred.bias'].mul_(0.02)
    sd[pre + 'cls_logits.weight'].mul_(0.25)
    A = sd[pre + 'cls_logits.bias'].numel()
    n_sizes = A // 3                                   # anchor a = ratio * n_sizes + size (rel_model_base.py:94-95: ratio-major)
    b = torch.full((A,), -8.0)
    b[0::n_sizes] = 8.0
    sd[pre + 'cls_logits.bias'].copy_(b)
    pre = 'detector.roi_heads.box_predictor.'
    g = torch.Generator().manual_seed(seed)
    sd[pre + 'cls_score.weight'].mul_(gain)
    sd[pre + 'cls_score.bias'].copy_(torch.randn(sd[pre + 'cls_score.bias'].shape, generator=g))
    sd[pre + 'bbox_pred.weight'].mul_(0.2)
    bb = sd[pre + 'bbox_pred.bias'].view(-1, 4)
    bb.mul_(0.2)
    bb[:, 2:] += 5.0 * math.log(grow)                  # [3P] RoIHeads box coder weights (10, 10, 5, 5)
    return sd


def synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=111, ragged=False, counts=None):
    """Blob-layout tuple (dataloaders/blob.py:244-249): (imgs list f32[3,S,S] on host, im_sizes, image_offset,
    gt_boxes f32[G,4], gt_classes i64[G,2], gt_rels i64[R,4], proposals, fns).  Boxes: x1,y1~U[0,0.68S),
    w,h~U[12,0.32S) clipped to S-1 (BASELINE.md 4 at S=592: U[0,400), U[12,192)).  counts: boxes per image (overrides n_boxes / ragged)."""
    rng = np.random.RandomState(seed)
    g = torch.Generator().manual_seed(seed)
    imgs = [torch.rand(3, S, S, generator=g) for _ in range(B)]
    boxes, classes, rels = [], [], []
    for b in range(B):
        n = int(counts[b]) if counts is not None else (n_boxes if not ragged else max(2, n_boxes - 3 * b))
        xy = rng.uniform(0, 400.0 / 592 * S, size=(n, 2))
        wh = rng.uniform(12, max(13.0, 192.0 / 592 * S), size=(n, 2))
        boxes.append(np.concatenate((xy, np.minimum(xy + wh, S - 1)), 1).astype(np.float32))
        classes.append(np.stack((np.full(n, b), rng.randint(1, 151, size=n)), 1).astype(np.int64))
        seen = set()
        while len(seen) < min(n_fg, n * (n - 1)):
            s, o = rng.randint(n), rng.randint(n)
            if s != o and (s, o) not in seen:
                seen.add((s, o))
                rels.append((b, s, o, rng.randint(1, 51)))
    im_sizes = np.array([[S, S, 1.0]] * B)
    return (imgs, im_sizes, 0, torch.from_numpy(np.concatenate(boxes)), torch.from_numpy(np.concatenate(classes)),
            torch.from_numpy(np.array(rels, dtype=np.int64)), None, ['synthetic%d' % b for b in range(B)])


def shard_batch(batch, lo, hi):
    """Images [lo, hi) of a Blob-layout tuple as a batch of their own (image ids renumbered from 0): the shard rank r of a
    data-parallel job holds when the global batch is `batch` (rank r takes images [r*B, (r+1)*B))."""
    imgs, im_sizes, off, boxes, classes, rels, props, fns = batch
    keep = (classes[:, 0] >= lo) & (classes[:, 0] < hi)
    cls = classes[keep].clone()
    cls[:, 0] -= lo
    rk = (rels[:, 0] >= lo) & (rels[:, 0] < hi)
    r = rels[rk].clone()
    r[:, 0] -= lo
    return (list(imgs[lo:hi]), im_sizes[lo:hi], off, boxes[keep].clone(), cls, r, props, list(fns[lo:hi]))


class GQASyntheticData(object):
    """`train_data` stand-in with GQA's vocabulary (BASELINE configs[4]): 1 703 object + 310 predicate classes + background
    (pretrain_detector.py:122 builds the GQA detector with 1 704 classes; the GQA scene graphs the reference's loader reads
    -- dataloaders/gqa.py -- carry 310 predicate names, i.e. 311 with the background)."""
    ind_to_classes = ['__background__'] + ['gqa_obj%d' % i for i in range(1, 1704)]
    ind_to_predicates = ['__background__'] + ['gqa_pred%d' % i for i in range(1, 311)]


def relabel_batch(batch, n_obj, n_pred, seed=0):
    """the same Blob tuple with object classes drawn from 1..n_obj-1 and predicates from 1..n_pred-1 (synthetic_batch draws VG's 150 / 50)"""
    rng = np.random.RandomState(seed)
    b = list(batch)
    cls = b[4].clone()
    cls[:, 1] = torch.from_numpy(rng.randint(1, n_obj, size=cls.shape[0]))
    rels = b[5].clone()
    rels[:, 3] = torch.from_numpy(rng.randint(1, n_pred, size=rels.shape[0]))
    b[4], b[5] = cls, rels
    return tuple(b)
