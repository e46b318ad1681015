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
