"""One feed variant on its own (for a kernel trace, or an A/B in one gpurun call):  python tools/feed_one.py host|hbm|host-d2d|host-chunked [steps]
  host-d2d      the real staging path with the H2D copy replaced by a device -> device copy of the same bytes (no PCIe traffic)
  host-chunked  the H2D copy as eight separate async copies"""
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

mode = sys.argv[1] if len(sys.argv) > 1 else 'host'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(dev).eval()
model.set_compute_dtype(torch.float16)
NB, w = 4, 8
hbs = []
for k in range(NB):
    hb = list(synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=111 + 1000 * k))
    hb[0] = [(im * 255).round().to(torch.uint8).permute(1, 2, 0).contiguous() for im in hb[0]]
    hbs.append(tuple(hb))


class D2D(DeviceStager):
    """experiment: the staged bytes come from a device-resident copy of the same packed batch (no PCIe traffic; events, waits and the worker
    thread as in the real path) -- 16 slots and 4 batches: slot k always holds batch k % 4"""

    def _launch(self, batch, slot, packed, ring=False, after=None):
        total = packed[2]
        if getattr(slot, 'devsrc', None) is None:
            slot.devsrc = slot.pinned[:total].to(self.device)
            torch.cuda.synchronize()
        real = slot.pinned
        slot.pinned = slot.devsrc          # copy_ below becomes device -> device
        try:
            return DeviceStager._launch(self, batch, slot, packed, ring, after)
        finally:
            slot.pinned = real




class Chunked(DeviceStager):
    """experiment: the same bytes as 8 separate async copies (is the cost of the transfer per copy or per byte?)"""

    def _launch(self, batch, slot, packed, ring=False, after=None):
        total = packed[2]
        real_copy = torch.Tensor.copy_

        def chunked(self_, src, non_blocking=False):
            n = self_.numel()
            step_ = (n // 8 + 255) // 256 * 256
            for o in range(0, n, step_):
                real_copy(self_[o:o + step_], src[o:o + step_], non_blocking=non_blocking)
            return self_
        torch.Tensor.copy_ = chunked
        try:
            return DeviceStager._launch(self, batch, slot, packed, ring, after)
        finally:
            torch.Tensor.copy_ = real_copy


st = {'host-d2d': D2D, 'host-chunked': Chunked}.get(mode, DeviceStager)(dev, slots=16)
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
feed = st.prefetch(hbs[i % NB] for i in range(n + w)) if mode.startswith('host') else (devb[i % NB] for i in range(n + w))
it = iter(feed)
for _ in range(w):
    tr.step(next(it))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    tr.step(next(it))
torch.cuda.synchronize()
print('%s: %.3f ms per step' % (mode, 1e3 * (time.perf_counter() - t0) / n))
