"""Does autograd keep the gradient tensors PredictFn.backward returns (AccumulateGrad steals them) or clone them?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sgg_amd  # noqa: E402
from sgg_amd import train as T  # noqa: E402
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch  # noqa: E402
from sgg_amd.trainer import Trainer  # noqa: E402

dev = 'cuda:0'
model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(dev)
model.set_compute_dtype(torch.bfloat16)
b = synthetic_batch()
batch = ([im.to(dev) for im in b[0]],) + tuple(t.to(dev) if isinstance(t, torch.Tensor) else t for t in b[1:])
ptrs = {}
orig = T.PredictFn.backward


def spy(ctx, *a):
    out = orig(ctx, *a)
    names = T.param_names(model)
    for n, g in zip(names, out[8:]):
        if g is not None:
            ptrs[n] = (g.data_ptr(), g.numel() * g.element_size(), g.dtype)
    return out


T.PredictFn.backward = staticmethod(spy)
tr = Trainer(model, lr=1e-3, pipeline=False)
tr.opt.step = lambda *a, **k: None            # keep .grad for inspection
tr.step(batch)
torch.cuda.synchronize()
named = dict(model.named_parameters())
cloned = 0
for n, (ptr, nb, dt) in ptrs.items():
    g = named[n].grad
    same = g is not None and g.data_ptr() == ptr
    if not same:
        cloned += nb
    print('%-34s %9.2f MB %-14s %s' % (n, nb / 1e6, str(dt), 'kept' if same else ('CLONED -> %s' % (g.dtype if g is not None else None))))
print('bytes cloned per step: %.1f MB' % (cloned / 1e6))
