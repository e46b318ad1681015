"""Which lines of this package still launch torch / runtime kernels inside a step?
Runs the bench workload (8 x 592^2, 32 boxes) for a few train and inference steps under torch.profiler with Python stacks, and lists
every device launch that is not one of the library's own entry points by (kernel name, innermost sgg_amd / bench frame): launches per
step and device microseconds per step.  `python tools/tiny_launches.py [train|infer] [steps]`"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgg_amd                                                              # noqa: E402
from sgg_amd.rel_model_base import to_device_with_mirror                   # noqa: E402
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch  # noqa: E402
from sgg_amd.trainer import Trainer                                         # noqa: E402

DEV = 'cuda:0'
mode = sys.argv[1] if len(sys.argv) > 1 else 'train'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4

model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(DEV)
batch = list(synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=111))
batch[0] = [im.to(DEV) for im in batch[0]]
batch[3], batch[4], batch[5] = batch[3].to(DEV), to_device_with_mirror(batch[4], DEV), to_device_with_mirror(batch[5], DEV)
batch = tuple(batch)

if mode == 'train':
    tr = Trainer(model, lr=1e-3, pipeline=True)

    def step():
        tr.step(batch)
    fin = tr.flush
else:
    model.eval()

    def step():
        with torch.no_grad():
            model([batch])
    fin = lambda: None                                                      # noqa: E731
for _ in range(4):
    step()
fin()
torch.cuda.synchronize()

import traceback                                                            # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode                 # noqa: E402

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NO_LAUNCH = ('empty', 'view', 'as_strided', 'slice', 'select', 'reshape', '_unsafe_view', 'expand', 'permute', 't.', 'transpose', 'detach',
             'alias', 'unsqueeze', 'squeeze', 'new_empty', 'is_pinned', 'lift_fresh', 'unbind', 'split', 'narrow', 'chunk', 'is_same_size',
             '_reshape_alias', 'set_', 'resize_', 'sym_', 'stride', 'size', 'numel', 'dim', 'is_', 'record_stream', 'unfold', 'flatten',
             'view_as', '_to_copy_placeholder')
sites = collections.defaultdict(int)


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__ if hasattr(func, '__name__') else str(func)
        full = str(func)
        short = full.replace('aten.', '')
        if not any(short.startswith(n) for n in NO_LAUNCH):
            on_dev = any(isinstance(a, torch.Tensor) and a.is_cuda for a in list(args) + list((kwargs or {}).values())
                         + [x for a in args if isinstance(a, (list, tuple)) for x in a])
            on_dev = on_dev or 'cuda' in str((kwargs or {}).get('device', ''))
            if on_dev or 'to' in short or 'tensor' in short:
                frames = [f for f in traceback.extract_stack() if root in f.filename and 'tiny_launches' not in f.filename]
                where = ' <- '.join('%s:%d' % (f.filename.replace(root + '/', ''), f.lineno) for f in reversed(frames[-3:]))
                sites[(short, where)] += 1
        return func(*args, **(kwargs or {}))


with torch.autograd.set_multithreading_enabled(False), Spy():
    for _ in range(steps):
        step()
    fin()
    torch.cuda.synchronize()
n = sum(sites.values())
print('%s: %d torch ops on device tensors over %d steps = %.1f per step' % (mode, n, steps, n / steps))
for (op, where), c in sorted(sites.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print('%5.1f x/step  %-28s %s' % (c / steps, op, where))
