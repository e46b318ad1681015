"""Experiment: the frozen detector's forward of batch k+1 on a second stream UNDER the head's forward / backward of batch k (the
detector reads no trainable weight and no result of step k).  Prints ms per step for the in-order step and for the look-ahead form.
python tools/exp_lookahead.py [steps]"""
import os
import sys
import time

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import sgg_amd
from sgg_amd.rel_model_base import to_device_with_mirror
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
from sgg_amd.trainer import Trainer


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    dev = 'cuda:0'
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(dev).eval()
    model.set_compute_dtype(torch.float16)
    batches = []
    for k in range(4):
        b = list(synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=111 + 1000 * k))
        b[0] = [im.to(dev) for im in b[0]]
        b[3], b[4], b[5] = b[3].to(dev), to_device_with_mirror(b[4], dev), to_device_with_mirror(b[5], dev)
        batches.append(tuple(b))
    tr = Trainer(model, lr=1e-3, pipeline=True)
    det = model.detector
    orig = det.features

    def run(n, ahead):
        side = torch.cuda.Stream(priority=0)
        pre = {}

        def features(images, dtype):
            key = id(images[0])
            if key in pre:
                out, ev = pre.pop(key)
                torch.cuda.current_stream().wait_event(ev)
                return out
            return orig(images, dtype)
        det.features = features if ahead else orig

        def prefetch(b):
            # (the planes of VGGDetector._bufs are shared: the look-ahead forward must own a second set -- here simply a second detector view)
            images = [b[0][i] for i in range(8)]
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                out = orig(images, torch.float16)
                ev = torch.cuda.Event()
                ev.record(side)
            pre[id(images[0])] = (out, ev)
        for i in range(3):
            tr.step(batches[i % 4])
        torch.cuda.synchronize()
        if ahead:
            prefetch(batches[0])
        t0 = time.perf_counter()
        for i in range(n):
            b = batches[i % 4]
            if ahead:
                prefetch(batches[(i + 1) % 4])     # queued BEFORE this step's head work: the side stream runs it beside that work
            tr.step(b)
        torch.cuda.synchronize()
        tr.flush()
        return (time.perf_counter() - t0) / n * 1e3
    for rep in range(2):
        print('in order   %.3f ms/step' % run(steps, False))
        print('look-ahead %.3f ms/step' % run(steps, True))


if __name__ == '__main__':
    main()
