"""Where does the host-input step lose time against the HBM-resident one?  Same trainer, same four batches, 100 steps per variant:
  hbm          batches resident in HBM
  host         DeviceStager.prefetch (pinned async copy on the copy stream, one step ahead)
  host-Nslots  ring depth"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch

import sgg_amd
from sgg_amd.blob import DeviceStager
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
from sgg_amd.trainer import Trainer

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(dev).eval()
model.set_compute_dtype(torch.float16)
NB = 4
hbs = []
for k in range(NB):
    hb = list(synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=111 + 1000 * k))
    hb[0] = [(im * 255).round().to(torch.uint8).permute(1, 2, 0).contiguous() for im in hb[0]]
    hbs.append(tuple(hb))


st = DeviceStager(dev)
devb = []
for hb in hbs:
    s_ = list(st.stage(hb))
    s_[0] = [im.clone() for im in s_[0]]
    for i in (3, 4, 5):
        m = getattr(s_[i], '_sgg_host', None)
        s_[i] = s_[i].clone()
        if m is not None:
            s_[i]._sgg_host = m
    devb.append(tuple(s_))
tr = Trainer(model, lr=1e-3, pipeline=True)
n, w = 100, 8


def run(name, feed):
    it = iter(feed)
    for _ in range(w):
        tr.step(next(it))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fs = 0.0
    for _ in range(n):
        a = time.perf_counter()
        b = next(it)
        fs += time.perf_counter() - a
        tr.step(b)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in it:
        pass
    print('%-12s %.3f ms per step (issued after %.3f ms per step, of which in the feed %.3f)' % (name, 1e3 * (t2 - t0) / n, 1e3 * (t1 - t0) / n, 1e3 * fs / n), flush=True)


for rep in range(2):
    run('host-3slots', DeviceStager(dev, slots=3).prefetch(hbs[i % NB] for i in range(n + w)))
    run('hbm', (devb[i % NB] for i in range(n + w)))
    run('host-16', DeviceStager(dev, slots=16).prefetch(hbs[i % NB] for i in range(n + w)))
    run('host-6slots', DeviceStager(dev, slots=6).prefetch(hbs[i % NB] for i in range(n + w)))
    run('host-3slots', DeviceStager(dev, slots=3).prefetch(hbs[i % NB] for i in range(n + w)))
    run('host-inline16', DeviceStager(dev, slots=16).prefetch((hbs[i % NB] for i in range(n + w)), threaded=False))
    run('host-inline3', DeviceStager(dev, slots=3).prefetch((hbs[i % NB] for i in range(n + w)), threaded=False))
