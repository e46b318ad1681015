"""50 pipelined train steps at the bench config on 4 alternating synthetic batches: loss must fall, no NaN/Inf anywhere."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgg_amd
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
from sgg_amd.trainer import Trainer
from sgg_amd.rel_model_base import to_device_with_mirror
dev = 'cuda:0'
model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(dev)
model.set_compute_dtype(torch.bfloat16)
batches = []
for s in range(4):
    b = list(synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=200 + s))
    b[0] = [im.to(dev) for im in b[0]]
    b[3], b[4], b[5] = b[3].to(dev), to_device_with_mirror(b[4], dev), to_device_with_mirror(b[5], dev)
    batches.append(tuple(b))
tr = Trainer(model, lr=1e-3, pipeline=True)
losses = []
for it in range(50):
    losses.append(tr.step(batches[it % 4]))
tr.flush()
torch.cuda.synchronize()
ls = [float(x) for x in losses]
print('loss first 4: %s  last 4: %s' % ([round(v, 3) for v in ls[:4]], [round(v, 3) for v in ls[-4:]]))
bad = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
print('non-finite parameters:', bad, ' grad norm (last step): %.3f' % tr.opt.grad_norm())
assert not bad and all(v == v for v in ls) and sum(ls[-4:]) < sum(ls[:4])
print('OK')
