"""Host side of a train step: how long the CPU needs to ISSUE one step (no synchronisation inside the loop) against the time the GPU
needs to run it.  If the first is well under the second the launch queue stays ahead and dispatch gaps cannot matter.
python tools/host_time.py [steps]"""
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
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    dev = 'cuda:0'
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(dev).eval()
    model.set_compute_dtype(torch.float16)
    b = list(synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=111))
    b[0] = [im.to(dev) for im in b[0]]
    b[3], b[4], b[5] = b[3].to(dev), to_device_with_mirror(b[4], dev), to_device_with_mirror(b[5], dev)
    b = tuple(b)
    tr = Trainer(model, lr=1e-3, pipeline=True)
    for _ in range(8):
        tr.step(b)
    torch.cuda.synchronize()
    issue = []
    t0 = time.perf_counter()
    for _ in range(steps):
        a = time.perf_counter()
        tr.step(b)
        issue.append(time.perf_counter() - a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    issue.sort()
    print('steps %d: issue loop %.3f ms per step (median call %.3f, p90 %.3f), drained after another %.2f ms -> %.3f ms per step on the GPU'
          % (steps, (t1 - t0) / steps * 1e3, issue[len(issue) // 2] * 1e3, issue[int(len(issue) * .9)] * 1e3, (t2 - t1) * 1e3,
             (t2 - t0) / steps * 1e3))
    # the same with a synchronisation after every step: the step's latency (issue + run, nothing hidden)
    lat = []
    for _ in range(20):
        a = time.perf_counter()
        tr.step(b)
        torch.cuda.synchronize()
        lat.append(time.perf_counter() - a)
    lat.sort()
    print('synchronised after every step: median %.3f ms' % (lat[len(lat) // 2] * 1e3))




def profile():
    """cProfile of the issue path (python tools/host_time.py profile): where the host's time per step goes"""
    import cProfile
    import pstats
    dev = 'cuda:0'
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(dev).eval()
    model.set_compute_dtype(torch.float16)
    b = list(synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=111))
    b[0] = [im.to(dev) for im in b[0]]
    b[3], b[4], b[5] = b[3].to(dev), to_device_with_mirror(b[4], dev), to_device_with_mirror(b[5], dev)
    b = tuple(b)
    if len(sys.argv) > 2 and sys.argv[2] == 'infer':
        def step_(b_):
            with torch.no_grad():
                return model([b_])
    else:
        tr = Trainer(model, lr=1e-3, pipeline=True)
        step_ = tr.step
    for _ in range(8):
        step_(b)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(10):
        step_(b)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats('tottime').print_stats(30)
    st.sort_stats('cumtime').print_stats(45)
    st.print_callees('_queue_update|train_weights|prepared')


if __name__ == '__main__':
    profile() if (len(sys.argv) > 1 and sys.argv[1] == 'profile') else main()
