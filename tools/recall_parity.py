"""R@K parity on a model that actually predicts something: overfit the head on a small fixed synthetic set (so that softmaxes are
confident instead of flat random-init ones), then evaluate the SAME images three ways -- HIP bf16, HIP fp32, CPU oracle fp32 -- through
the evaluator and compare R@20/50/100 (GC and no-GC), sgcls and predcls.  The north star asks for R@50 within +-0.1.
    python tools/recall_parity.py [train_steps] [images]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sgg_oracle as O
import sgg_amd
from sgg_amd.recall import BasicSceneGraphEvaluator
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
from sgg_amd.trainer import Trainer
from sgg_amd.rel_model_base import to_device_with_mirror
dev = 'cuda:0'
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 16
S, NB = 592, 16
model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(dev)
model.set_compute_dtype(torch.bfloat16)
model.dropout_p = 0.0                      # memorise: the point is a confident model, not generalisation
B = 8
# Images with something to learn from (uniform noise through a random frozen VGG gives every box the same features): each box is painted
# with its class's colour on a grey canvas, and the predicate of a relation is a function of the two classes.
NCLS = int(os.environ.get('NCLS', '24'))
def colour(c):
    return torch.tensor([(c * 37) % 256, (c * 91) % 256, (c * 53) % 256], dtype=torch.float32) / 255.0
def make_batch(seed):
    rng = np.random.RandomState(seed)
    imgs, boxes, classes, rels = [], [], [], []
    for b in range(B):
        img = torch.full((3, S, S), 0.5) + 0.02 * torch.from_numpy(rng.randn(3, S, S).astype(np.float32))
        xy = rng.uniform(0, S - 130, size=(NB, 2))
        wh = rng.uniform(50, 125, size=(NB, 2))
        bx = np.concatenate((xy, xy + wh), 1).astype(np.float32)
        cls = rng.randint(1, NCLS + 1, size=NB)
        for k in range(NB):
            x1, y1, x2, y2 = [int(v) for v in bx[k]]
            img[:, y1:y2, x1:x2] = colour(int(cls[k]))[:, None, None]
        seen = set()
        while len(seen) < 8:
            s_, o_ = rng.randint(NB), rng.randint(NB)
            if s_ != o_ and (s_, o_) not in seen:
                seen.add((s_, o_))
                rels.append((b, s_, o_, 1 + (int(cls[s_]) * 7 + int(cls[o_]) * 3) % 50))
        imgs.append(img.clamp(0, 1)); boxes.append(bx)
        classes.append(np.stack((np.full(NB, b), cls), 1).astype(np.int64))
    return [imgs, np.array([[S, S, 1.0]] * B), 0, torch.from_numpy(np.concatenate(boxes)), torch.from_numpy(np.concatenate(classes)),
            torch.from_numpy(np.array(rels, dtype=np.int64)), None, ['img%d_%d' % (seed, b) for b in range(B)]]
batches = [make_batch(900 + s) for s in range(n_img // B)]
def on_dev(b):
    b = list(b)
    b[0] = [im.to(dev) for im in b[0]]
    b[3], b[4], b[5] = b[3].to(dev), to_device_with_mirror(b[4], dev), to_device_with_mirror(b[5], dev)
    return tuple(b)
dbatches = [on_dev(b) for b in batches]
tr = Trainer(model, lr=float(os.environ.get('LR', '0.02')), pipeline=True)
t0 = time.time()
for it in range(steps):
    loss = tr.step(dbatches[it % len(dbatches)])
    if it % 50 == 0 or it == steps - 1:
        print('step %4d loss %.4f' % (it, float(loss)), flush=True)
tr.flush()
torch.cuda.synchronize()
print('trained %d steps in %.1f s' % (steps, time.time() - t0))
sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
model.eval()
KS = (20, 50, 100)
def single(b, i):
    """image i of batch b as a one-image batch (the reference evaluates one image per call)"""
    sel = (b[4][:, 0] == i)
    cls = b[4][sel].clone(); cls[:, 0] = 0
    rel = b[5][b[5][:, 0] == i].clone(); rel[:, 0] = 0
    return ([b[0][i]], b[1][i:i + 1], 0, b[3][sel].clone(), cls, rel, None, [b[7][i]])
held = [make_batch(5000 + s) for s in range(n_img // B)]      # same colour -> class rule, images the head has not seen
results = {}
for split, bset in (('train', batches), ('held-out', held)):
  for mode in ('sgcls', 'predcls'):
    model.mode = mode
    evs = {(w, mp): BasicSceneGraphEvaluator(mode, multiple_preds=mp) for w in ('hip_bf16', 'hip_fp32', 'oracle_fp32') for mp in (False, True)}
    for b in bset:
        for i in range(B):
            one = single(b, i)
            gt = {'gt_boxes': one[3].numpy(), 'gt_classes': one[4][:, 1].numpy(), 'gt_relations': one[5][:, 1:].numpy()}
            outs = {}
            with torch.no_grad():
                for name, dt in (('hip_bf16', torch.bfloat16), ('hip_fp32', torch.float32)):
                    model.set_compute_dtype(dt)
                    outs[name] = model([one])
                ref = O.forward_gtbox(one[0], one[3], one[4], one[5], sd, mode=mode)
                outs['oracle_fp32'] = ref['dets']
            for name, (boxes, objs, scores, rels, pscores) in outs.items():
                pr = {'pred_boxes': boxes, 'pred_classes': objs, 'pred_rel_inds': rels, 'obj_scores': scores, 'rel_scores': pscores}
                for mp in (False, True):
                    evs[(name, mp)].evaluate_scene_graph_entry(gt, pr)
    for (name, mp), ev in evs.items():
        results[(split, mode, name, mp)] = {k: 100 * float(np.mean(ev.result_dict[mode + '_recall'][k])) for k in KS}
print()
print('%-9s %-8s %-12s %-5s ' % ('images', 'mode', 'path', 'GC') + ' '.join('R@%-6d' % k for k in KS))
for (split, mode, name, mp), r in results.items():
    print('%-9s %-8s %-12s %-5s ' % (split, mode, name, 'no' if mp else 'yes') + ' '.join('%-8.2f' % r[k] for k in KS))
worst = max(abs(results[(sp, m, a, mp)][k] - results[(sp, m, 'oracle_fp32', mp)][k]) for sp in ('train', 'held-out') for m in ('sgcls', 'predcls')
            for a in ('hip_bf16', 'hip_fp32') for mp in (False, True) for k in KS)
print('largest |R@K(HIP) - R@K(oracle)| over train / held-out, both modes, GC / no-GC, K in %s: %.2f points (%d GT triplets per set)'
      % (KS, worst, 8 * n_img))
