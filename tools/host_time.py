"""Host (Python + ctypes launch path) time per train / inference step vs the GPU-bound wall time."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgg_amd
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
from sgg_amd.trainer import Trainer
dev = 'cuda:0'
model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(dev)
model.set_compute_dtype(torch.float16)
b = list(synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=111))
b[0] = [im.to(dev) for im in b[0]]
from sgg_amd.rel_model_base import to_device_with_mirror
b[3], b[4], b[5] = b[3].to(dev), to_device_with_mirror(b[4], dev), to_device_with_mirror(b[5], dev)
b = tuple(b)
tr = Trainer(model, lr=1e-3, pipeline=True)
for _ in range(5):
    tr.step(b)
torch.cuda.synchronize()
host, wall = [], []
for _ in range(20):
    t0 = time.perf_counter()
    tr.step(b)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(t1 - t0); wall.append(t2 - t0)
print('train: host %.2f ms (min %.2f)  host+drain %.2f ms' % (1e3 * sum(host) / len(host), 1e3 * min(host), 1e3 * sum(wall) / len(wall)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    tr.step(b)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
