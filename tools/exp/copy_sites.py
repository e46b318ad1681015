"""Which lines of sgg_amd make torch move bytes (copy_ / clone / contiguous / cat / to / zero_ / fill_) in one train step: the Python-level
calls are wrapped and logged with their byte counts and the innermost sgg_amd frame."""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sgg_amd  # noqa: E402
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch  # noqa: E402
from sgg_amd.trainer import Trainer  # noqa: E402

dev = 'cuda:0'
model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(dev)
model.set_compute_dtype(torch.bfloat16)
from sgg_amd.rel_model_base import to_device_with_mirror  # noqa: E402
b = list(synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=111))
b[0] = [im.to(dev) for im in b[0]]
b[3], b[4], b[5] = b[3].to(dev), to_device_with_mirror(b[4], dev), to_device_with_mirror(b[5], dev)
batch = tuple(b)
tr = Trainer(model, lr=1e-3, pipeline=True)
INFER = len(sys.argv) > 1 and sys.argv[1] == 'infer'
if INFER:
    model.eval()
run = (lambda: model([batch])) if INFER else (lambda: tr.step(batch))
for _ in range(4):
    run()
torch.cuda.synchronize()
log = collections.defaultdict(lambda: [0, 0])
on = [False]


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if 'sgg_amd/' in fr.filename and 'copy_sites' not in fr.filename:
            return '%s:%d' % (fr.filename.split('sgg_amd/')[-1], fr.lineno)
    return 'autograd / other'


def wrap(owner, name, size_of):
    orig = getattr(owner, name)

    def f(*a, **k):
        out = orig(*a, **k)
        if on[0]:
            try:
                nb = size_of(a, k, out)
            except Exception:
                nb = 0
            if nb:
                e = log[(name, site())]
                e[0] += 1
                e[1] += nb
        return out
    setattr(owner, name, f)


nbytes = lambda t: t.numel() * t.element_size() if isinstance(t, torch.Tensor) and t.is_cuda else 0   # noqa: E731
wrap(torch.Tensor, 'copy_', lambda a, k, o: nbytes(a[0]))
wrap(torch.Tensor, 'clone', lambda a, k, o: nbytes(o))
wrap(torch.Tensor, 'contiguous', lambda a, k, o: nbytes(o) if o.data_ptr() != a[0].data_ptr() else 0)
wrap(torch.Tensor, 'to', lambda a, k, o: nbytes(o) if isinstance(o, torch.Tensor) and (not a[0].is_cuda or o.data_ptr() != a[0].data_ptr()) else 0)
wrap(torch.Tensor, 'float', lambda a, k, o: nbytes(o) if o.data_ptr() != a[0].data_ptr() else 0)
wrap(torch.Tensor, 'zero_', lambda a, k, o: nbytes(a[0]))
wrap(torch.Tensor, 'fill_', lambda a, k, o: nbytes(a[0]))
wrap(torch, 'cat', lambda a, k, o: nbytes(o))
wrap(torch, 'zeros', lambda a, k, o: nbytes(o))
wrap(torch, 'zeros_like', lambda a, k, o: nbytes(o))
wrap(torch, 'ones', lambda a, k, o: nbytes(o))
wrap(torch, 'full', lambda a, k, o: nbytes(o))
wrap(torch, 'tensor', lambda a, k, o: nbytes(o))
wrap(torch, 'arange', lambda a, k, o: nbytes(o))
wrap(torch.Tensor, 'long', lambda a, k, o: nbytes(o) if o.data_ptr() != a[0].data_ptr() else 0)
wrap(torch.Tensor, 'int', lambda a, k, o: nbytes(o) if o.data_ptr() != a[0].data_ptr() else 0)
wrap(torch.Tensor, 'add_', lambda a, k, o: nbytes(a[0]))
wrap(torch.Tensor, '__getitem__', lambda a, k, o: nbytes(o) if isinstance(o, torch.Tensor) and o._base is None and o.data_ptr() != a[0].data_ptr() else 0)
on[0] = True
for _ in range(2):
    run()
on[0] = False
torch.cuda.synchronize()
rows = sorted(log.items(), key=lambda kv: -kv[1][1])
print('per step: %.1f calls, %.1f MB' % (sum(v[0] for _, v in rows) / 2, sum(v[1] for _, v in rows) / 2e6))
for (name, where), (n, nb) in rows[:70]:
    print('%5.1f/step %9.2f MB/step  %-11s %s' % (n / 2, nb / 2e6, name, where))
