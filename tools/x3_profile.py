"""Per-call time of a train step / inference forward in a given compute mode (f16 | x3 | f32): which launches the x3 mode spends its time in.
python tools/x3_profile.py [mode] [train|infer]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch

import sgg_amd
from sgg_amd import _lib
from sgg_amd.rel_model_base import to_device_with_mirror
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
from sgg_amd.trainer import Trainer


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else 'x3'
    what = sys.argv[2] if len(sys.argv) > 2 else 'train'
    dev = 'cuda:0'
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(dev).eval()
    if mode == 'x3':
        model.set_compute_dtype(torch.float32, split3=True)
    else:
        model.set_compute_dtype({'f16': torch.float16, 'f32': torch.float32}[mode])
    b = list(synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=111))
    b[0] = [im.to(dev) for im in b[0]]
    b[3], b[4], b[5] = b[3].to(dev), to_device_with_mirror(b[4], dev), to_device_with_mirror(b[5], dev)
    b = tuple(b)
    tr = Trainer(model, lr=1e-3)

    def step():
        if what == 'train':
            tr.step(b)
        else:
            model.eval()
            with torch.no_grad():
                model([b])
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    prof = {}
    _lib.profiler = prof
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    _lib.profiler = None
    rows = sorted(((sum(a.elapsed_time(c) for a, c in evs) / 3, n, t, len(evs) // 3) for (n, t), evs in prof.items()), reverse=True)
    total = sum(r[0] for r in rows)
    print('%s %s: sum of call times %.2f ms per step' % (mode, what, total))
    for ms, n, t, k in rows[:24]:
        print('  %8.3f ms  %3d x  %-26s %s' % (ms, k, n, t))


if __name__ == '__main__':
    main()
