"""Which torch (aten) operators launch GPU kernels inside one train step, and from which line of sgg_amd: torch.profiler with stacks."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sgg_amd  # noqa: E402
from sgg_amd.rel_model_base import to_device_with_mirror  # noqa: E402
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch  # noqa: E402
from sgg_amd.trainer import Trainer  # noqa: E402

dev = 'cuda:0'
mode = sys.argv[1] if len(sys.argv) > 1 else 'train'
model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(dev)
model.set_compute_dtype(torch.bfloat16)
b = list(synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=111))
b[0] = [im.to(dev) for im in b[0]]
b[3], b[4], b[5] = b[3].to(dev), to_device_with_mirror(b[4], dev), to_device_with_mirror(b[5], dev)
b = tuple(b)
tr = Trainer(model, lr=1e-3)
if mode == 'train':
    step = lambda: tr.step(b)      # noqa: E731
else:
    model.eval()
    step = lambda: model([b])      # noqa: E731
for _ in range(5):
    step()
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.key_averages(group_by_stack_n=25):
    if not ev.key.startswith('aten::') or ev.device_time_total <= 0:
        continue
    where = 'other'
    for fr in ev.stack or []:
        if 'sgg_amd/' in fr:
            where = fr.split('sgg_amd/')[-1][:70]
            break
    e = agg[(ev.key, where)]
    e[0] += ev.count
    e[1] += ev.self_device_time_total
tot = 0.0
for (name, where), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if us <= 0:
        continue
    tot += us / N
    print('%7.1f us/step  %5.1f calls/step  %-28s %s' % (us / N, n / N, name, where))
print('total %.1f us per step in aten operators (self device time)' % tot)
