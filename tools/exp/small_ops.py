"""Which lines of sgg_amd launch torch's own small kernels (fills, copies, casts, cats) in one train step: torch.profiler with
stacks, grouped by the innermost sgg_amd frame.  python tools/exp/small_ops.py [train|infer]"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sgg_amd  # noqa: E402
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch  # noqa: E402
from sgg_amd.trainer import Trainer  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'train'
dev = 'cuda:0'
model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(dev)
model.set_compute_dtype(torch.bfloat16)
batch = tuple(t.to(dev) if isinstance(t, torch.Tensor) else t for t in synthetic_batch())
batch = (([im.to(dev) for im in batch[0]],) + batch[1:])
if mode == 'train':
    tr = Trainer(model, lr=1e-3, pipeline=True)
    step = lambda: tr.step(batch)  # noqa: E731
else:
    model.eval()
    step = lambda: model([batch])  # noqa: E731
for _ in range(4):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_time_total <= 0 or not ev.name.startswith('aten::') or ev.cpu_parent is not None and ev.cpu_parent.name.startswith('aten::'):
        continue
    site = 'unknown'
    for fr in ev.stack:
        if 'sgg_amd/' in fr or 'bench.py' in fr:
            site = fr.split('sgg_amd/')[-1] if 'sgg_amd/' in fr else fr
            break
    a = agg[(ev.name, site)]
    a[0] += 1
    a[1] += ev.device_time_total
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print('%d distinct (op, site); per step: %.1f ops, %.1f us of device time' % (len(rows), sum(v[0] for _, v in rows) / 3, sum(v[1] for _, v in rows) / 3))
for (name, site), (n, us) in rows[:45]:
    print('%5.1f/step %8.1f us/step  %-28s %s' % (n / 3, us / 3, name, site[:110]))
