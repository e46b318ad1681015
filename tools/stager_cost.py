"""Host cost of the staging path alone: iterate DeviceStager.prefetch over the bench's four host batches without running a step
(threaded and in-line), main-thread time per batch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sgg_amd.blob import DeviceStager
from sgg_amd.synthetic import synthetic_batch

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
hbs = []
for k in range(4):
    hb = list(synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=111 + 1000 * k))
    hb[0] = [(im * 255).round().to(torch.uint8).permute(1, 2, 0).contiguous() for im in hb[0]]
    hbs.append(tuple(hb))
st = DeviceStager(dev)
for threaded in (True, False):
    for sleep_ms in (0.0, 5.0):
        n = 100
        it = st.prefetch((hbs[i % 4] for i in range(n + 5)), threaded=threaded)
        for _ in range(5):
            next(it)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        busy = 0.0
        for _ in range(n):
            a = time.perf_counter()
            next(it)
            busy += time.perf_counter() - a
            if sleep_ms:
                time.sleep(sleep_ms / 1e3)      # stands in for the step's issue time: the worker packs meanwhile
        torch.cuda.synchronize()
        for _ in it:
            pass
        print('threaded=%s, %.0f ms between batches: main thread %.3f ms per batch inside the generator' % (threaded, sleep_ms, 1e3 * busy / n))
