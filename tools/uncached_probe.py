"""Does the update's stream hurt the VGG forward through the CACHES (L2 / MALL pollution) or through the memory fabric itself?  The fp32
masters and momentum buffers of the two fc6 weights (86 % of the update's bytes, touched by nothing but the update kernel) are moved to
UNCACHED device memory (hipExtMallocWithFlags(hipDeviceMallocUncached)) before the step's graphs are captured; the replayed step is timed
with and without.     python tools/uncached_probe.py [0|1]        (GPU)"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import sgg_amd
from sgg_amd.rel_model_base import to_device_with_mirror
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
from sgg_amd.trainer import Trainer

DEV = 'cuda:0'


class Holder(object):
    """device memory from hipExtMallocWithFlags behind __cuda_array_interface__ (kept alive by the tensor made from it)"""

    def __init__(self, hip, nbytes, flags, shape):
        p = ctypes.c_void_p()
        rc = hip.hipExtMallocWithFlags(ctypes.byref(p), ctypes.c_size_t(nbytes), ctypes.c_uint(flags))
        if rc != 0 or not p.value:
            raise RuntimeError('hipExtMallocWithFlags -> %d' % rc)
        self.ptr = p.value
        self.__cuda_array_interface__ = {'shape': tuple(shape), 'typestr': '<f4', 'data': (self.ptr, False), 'version': 2}


def uncached_like(hip, t):
    h = Holder(hip, t.numel() * 4, 0x3, t.shape)
    out = torch.as_tensor(h, device=DEV)
    out._holder = h
    out.copy_(t)
    return out


def main():
    on = (sys.argv[1] if len(sys.argv) > 1 else '1') == '1'
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(DEV).eval()
    model.set_compute_dtype(torch.float16)
    b = list(synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=111))
    b[0] = [(im * 255).round().to(torch.uint8).permute(1, 2, 0).contiguous().to(DEV) for im in b[0]]
    b[3], b[4], b[5] = b[3].to(DEV), to_device_with_mirror(b[4], DEV), to_device_with_mirror(b[5], DEV)
    b = tuple(b)
    tr = Trainer(model, lr=1e-3, pipeline=True)
    tr.step(b)
    tr.flush()
    torch.cuda.synchronize()
    if on:
        hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamdhip64.so'))
        moved = 0
        for n, p in model.named_parameters():
            if p.requires_grad and p.numel() > 50_000_000:
                st = tr.opt.state[p]
                st['momentum_buffer'] = uncached_like(hip, st['momentum_buffer'])
                p.data = uncached_like(hip, p.data)
                moved += 2 * p.numel() * 4
                print('uncached:', n, tuple(p.shape))
        print('moved %.2f GB to uncached memory' % (moved / 1e9))
    for _ in range(12):
        tr.step(b)
    torch.cuda.synchronize()
    g = tr.graphs
    print('graphs:', g.stats, g.disabled)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(80):
            tr.step(b)
        torch.cuda.synchronize()
        print('uncached=%d: %.3f ms per step' % (on, (time.perf_counter() - t0) / 80 * 1e3))
    u = next(iter(g.U.values()))[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        u.replay()
    torch.cuda.synchronize()
    print('update graph alone: %.3f ms' % ((time.perf_counter() - t0) / 30 * 1e3))


if __name__ == '__main__':
    main()
