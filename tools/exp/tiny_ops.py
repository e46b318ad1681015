"""Which lines of sgg_amd issue the small torch kernels (fills, copies, casts, cats) of a train step."""
import collections, os, sys, torch
sys.path.insert(0, '.')
import sgg_amd
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
from sgg_amd.trainer import Trainer
from sgg_amd.rel_model_base import to_device_with_mirror
from torch.profiler import profile, ProfilerActivity
dev = 'cuda:0'
model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(dev)
model.set_compute_dtype(torch.bfloat16)
b = list(synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=111))
b[0] = [im.to(dev) for im in b[0]]
b[3], b[4], b[5] = b[3].to(dev), to_device_with_mirror(b[4], dev), to_device_with_mirror(b[5], dev)
b = tuple(b)
tr = Trainer(model, lr=1e-3, pipeline=True)
for _ in range(4):
    tr.step(b)
torch.cuda.synchronize()
N = 3
import traceback
cnt = collections.Counter()
def wrap(owner, name):
    orig = getattr(owner, name)
    def f(*a, **k):
        for fr in reversed(traceback.extract_stack(limit=8)[:-1]):
            if '/sgg_amd/' in fr.filename:
                r = orig(*a, **k)
                t = r if torch.is_tensor(r) else (a[0] if a and torch.is_tensor(a[0]) else None)
                if t is not None and t.is_cuda or name in ('tensor', 'to'):
                    cnt[(name, '%s:%d %s' % (fr.filename.split('/sgg_amd/')[-1], fr.lineno, (fr.line or '')[:70]))] += 1
                return r
        return orig(*a, **k)
    setattr(owner, name, f)
for nm in ('copy_', 'zero_', 'fill_', 'to', 'float', 'contiguous', 'clone', 'long', 'new_zeros', 'div_', 'bfloat16'):
    wrap(torch.Tensor, nm)
for nm in ('cat', 'zeros', 'tensor', 'ones', 'zeros_like', 'stack', 'full', 'arange'):
    wrap(torch, nm)
for _ in range(N):
    tr.step(b)
torch.cuda.synchronize()
for (name, loc), c in cnt.most_common(80):
    print('%5.1f /step  %-11s %s' % (c / N, name, loc))
