"""N steps of the bench workload in one compute mode, nothing else (to be run under rocprofv3 --kernel-trace):
    python tools/mode_steps.py [f16|bf16|x3|f32] [train|infer] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch

import sgg_amd
from sgg_amd.rel_model_base import to_device_with_mirror
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
from sgg_amd.trainer import Trainer


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else 'x3'
    what = sys.argv[2] if len(sys.argv) > 2 else 'train'
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    dev = 'cuda:0'
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(dev).eval()
    if mode == 'x3':
        model.set_compute_dtype(torch.float32, split3=True)
    else:
        model.set_compute_dtype({'f16': torch.float16, 'bf16': torch.bfloat16, 'f32': torch.float32}[mode])
    b = list(synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=111))
    b[0] = [im.to(dev) for im in b[0]]
    b[3], b[4], b[5] = b[3].to(dev), to_device_with_mirror(b[4], dev), to_device_with_mirror(b[5], dev)
    b = tuple(b)
    tr = Trainer(model, lr=1e-3, graph=False) if what == 'train' else None

    def step():
        if what == 'train':
            tr.step(b)
        else:
            model.eval()
            with torch.no_grad():
                model([b])
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if tr is not None:
        tr.flush()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print('%s %s: %.3f ms per step, %.1f images/s' % (mode, what, 1e3 * dt, 8 / dt))


if __name__ == '__main__':
    main()
