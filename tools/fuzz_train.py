"""Randomised gradient check of the whole head (fp32, dropout off): random ragged batches -- different boxes per image, duplicate and
missing relations -- HIP forward + backward against torch autograd of the oracle's predict() for all 40 trainable tensors.
    python tools/fuzz_train.py [cases]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sgg_oracle as O
import sgg_amd
from sgg_amd.synthetic import SyntheticData, init_weights
from sgg_amd.train import param_names
dev = 'cuda:0'
S = 160
model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls', min_size=S, max_size=S))
sd = {k: v.clone() for k, v in model.state_dict().items()}
model.to(dev)
model.set_compute_dtype(torch.float32)
model.dropout_p = 0.0
rng = np.random.RandomState(int(os.environ.get('FUZZ_SEED', '0')))
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 4
t0 = time.time()
worst_all = 0.0
explained = 0
for case in range(cases):
    B = int(rng.randint(1, 4))
    g = torch.Generator().manual_seed(int(rng.randint(1 << 30)))
    imgs = [torch.rand(3, S, S, generator=g) for _ in range(B)]
    boxes, classes, rels = [], [], []
    for b in range(B):
        n = int(rng.randint(2, 10))
        xy = rng.uniform(0, 100, size=(n, 2))
        boxes.append(np.concatenate((xy, np.minimum(xy + rng.uniform(8, 60, size=(n, 2)), S - 1)), 1).astype(np.float32))
        classes.append(np.stack((np.full(n, b), rng.randint(1, 151, size=n)), 1).astype(np.int64))
        for _ in range(int(rng.randint(0, 5))):
            s, o = rng.randint(n), rng.randint(n)
            if s != o:
                rels.append((b, s, o, rng.randint(1, 51)))           # duplicates allowed
    if not rels:
        rels.append((0, 0, 1, 3))
    batch = (imgs, np.array([[S, S, 1.0]] * B), 0, torch.from_numpy(np.concatenate(boxes)), torch.from_numpy(np.concatenate(classes)),
             torch.from_numpy(np.array(rels, dtype=np.int64)), None, ['f%d' % b for b in range(B)])
    model.load_state_dict(sd)
    model.train()
    res = model([batch])
    Wo, Wr = torch.randn(res.rm_obj_dists.shape, generator=g), torch.randn(res.rel_dists.shape, generator=g)
    model.zero_grad()
    ((res.rm_obj_dists * Wo.to(dev)).sum() + (res.rel_dists * Wr.to(dev)).sum()).backward()
    pn = set(param_names(model))
    p = {k: v.clone().requires_grad_(k in pn) for k, v in sd.items()}
    od, rd = O.predict(res.node_feat.float().cpu().contiguous(), res.edge_feat.float().cpu().contiguous(), res.rel_inds.cpu().numpy(),
                       res.rois.cpu().numpy(), p, training=True)
    torch.testing.assert_close(res.rel_dists.detach().cpu(), rd.detach(), atol=1e-3, rtol=1e-3)
    ((od * Wo).sum() + (rd * Wr).sum()).backward()
    named = dict(model.named_parameters())
    worst = 0.0
    bad = False
    for n_ in param_names(model):
        ref, got = p[n_].grad, named[n_].grad.cpu()
        err = float((got - ref).abs().max()) / (float(ref.abs().max()) + 1e-6)
        worst = max(worst, err)
        if err >= 3e-3:
            d = (got - ref).abs()
            idx = np.unravel_index(int(d.argmax()), tuple(d.shape))
            print('MISMATCH case', case, n_, err, 'B', B, 'nodes', res.rm_obj_dists.shape[0], 'edges', res.rel_dists.shape[0], 'rels', rels,
                  'at', idx, float(got[idx]), float(ref[idx]), 'n bad', int((d > 3e-3 * float(ref.abs().max())).sum()), flush=True)
            bad = True
    if not bad:
        worst_all = max(worst_all, worst)
    if bad:
        # conditioning of the two batch-statistic BatchNorms: a channel whose ReLU output is (almost) constant over the batch has
        # variance ~0, invstd ~ 1/sqrt(eps) = 316, and amplifies summation-order differences in its own and upstream gradients
        import torch.nn.functional as F
        roi, ri = res.rois.cpu(), res.rel_inds.cpu()
        pr = torch.cat((roi[:, 1:][ri[:, 1]], roi[:, 1:][ri[:, 2]]), 1).numpy()
        rects = torch.from_numpy(O.draw_union_boxes(pr, 27) - np.float32(0.5))
        h1 = F.relu(F.conv2d(rects, sd['union_boxes.conv.0.weight'], sd['union_boxes.conv.0.bias'], stride=16, padding=3))
        var1 = h1.double().var(dim=(0, 2, 3), unbiased=False)
        x = F.batch_norm(h1, None, None, sd['union_boxes.conv.2.weight'], sd['union_boxes.conv.2.bias'], True, 0.01, 1e-5)
        h3 = F.relu(F.conv2d(F.max_pool2d(x, 3, 2, 1), sd['union_boxes.conv.4.weight'], sd['union_boxes.conv.4.bias'], stride=16, padding=1))
        var2 = h3.double().var(dim=(0, 2, 3), unbiased=False)
        print('boxes per image', [len(b_) for b_ in boxes], '| BN1 channels with var < 1e-4: %d, BN2: %d' % (int((var1 < 1e-4).sum()), int((var2 < 1e-4).sum())))
        # ReLU kinks: a pre-activation within rounding distance of 0 may get the other sign on the other implementation (K = 25088
        # fp32 terms summed in a different order), which switches one unit's gradient on or off -- a discontinuity, not an error
        with torch.no_grad():
            up = {k[len('union_boxes.'):]: v for k, v in sd.items() if k.startswith('union_boxes.')}
            efe = O.union_boxes_and_feats(res.edge_feat.float().cpu(), res.rois.cpu().numpy(), res.rel_inds.cpu().numpy()[:, 1:], up, training=True)
            z6 = efe.reshape(efe.shape[0], -1) @ sd['roi_fmap.1.0.weight'].t() + sd['roi_fmap.1.0.bias']
            zn = res.node_feat.float().cpu().reshape(res.node_feat.shape[0], -1) @ sd['roi_fmap_obj.0.weight'].t() + sd['roi_fmap_obj.0.bias']
            # ... and the second Linear of each branch (its ReLU feeds the next layer in the node branch; TwoMLPHead copies also in the edge branch)
            z7n = torch.relu(zn) @ sd['roi_fmap_obj.3.weight'].t() + sd['roi_fmap_obj.3.bias']
            z7e = torch.relu(z6) @ sd['roi_fmap.1.3.weight'].t() + sd['roi_fmap.1.3.bias']
        kinks = int((z6.abs() < 2e-5).sum()) + int((zn.abs() < 2e-5).sum()) + int((z7n.abs() < 2e-5).sum())
        print('  fc7 pre-activations within 2e-5 of zero: nodes %d (min |z| %.1e), edges %d (min |z| %.1e; no ReLU after it in the VGG head)' % (
            int((z7n.abs() < 2e-5).sum()), float(z7n.abs().min()), int((z7e.abs() < 2e-5).sum()), float(z7e.abs().min())))
        print('  fc6 pre-activations within 2e-5 of zero: edges %d (min |z| %.1e), nodes %d (min |z| %.1e)' % (
            int((z6.abs() < 2e-5).sum()), float(z6.abs().min()), int((zn.abs() < 2e-5).sum()), float(zn.abs().min())))
        # which of the 4 positions the max-pool picks: an arg-max flip between two nearly equal values is a legitimate discontinuity
        from sgg_amd import ops
        pooled, idx = F.max_pool2d(x, 3, 2, 1, return_indices=True)
        rows = h1.permute(0, 2, 3, 1).reshape(-1, h1.shape[1]).contiguous().to(dev)
        _, arg, _, _ = ops.bn_train(rows, sd['union_boxes.conv.2.weight'].to(dev), sd['union_boxes.conv.2.bias'].to(dev),
                                    torch.zeros(h1.shape[1], device=dev), torch.ones(h1.shape[1], device=dev), 1e-5, 0.01, True)
        t_idx = idx.reshape(idx.shape[0], idx.shape[1])
        flips = (arg.cpu().long() != t_idx)
        xs = x.reshape(x.shape[0], x.shape[1], 4)
        top2 = xs.topk(2, dim=2).values
        gap = (top2[..., 0] - top2[..., 1])
        print('  arg-max flips vs torch: %d of %d; gap between the two largest values at the flips: max %.2e' % (
            int(flips.sum()), flips.numel(), float(gap[flips].max()) if flips.any() else 0.0))
        for n_ in ('union_boxes.conv.2.weight', 'union_boxes.conv.6.weight'):
            ref, got = p[n_].grad, named[n_].grad.cpu()
            d = (got - ref).abs()
            worst_ch = torch.argsort(d, descending=True)[:5]
            vv = var1 if '.2.' in n_ else var2
            print('  %s: worst channels %s  their batch variance %s' % (n_, worst_ch.tolist(), ['%.1e' % float(vv[c]) for c in worst_ch]))
        if kinks == 0 and not flips.any():
            raise AssertionError('case %d: gradient mismatch without a ReLU kink or an arg-max flip to explain it' % case)
        explained += 1
    print('case %d: B=%d nodes=%d edges=%d rels=%d  worst relative gradient error %.2e' % (case, B, res.rm_obj_dists.shape[0],
                                                                                              res.rel_dists.shape[0], len(rels), worst), flush=True)
print('fuzz_train ok: %d cases (%d explained by a ReLU kink), worst %.2e, %.0f s' % (cases, explained, worst_all, time.time() - t0))
