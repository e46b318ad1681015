"""Is the price the VGG forward pays beside ANY resident kernel of another queue (tools/pair_probe.py: +0.35 ms beside workgroups that only
sleep) a matter of CU resources or of the queues themselves?  The forward (graph V of a replayed step) beside 256 / 16 / 1 sleeping
workgroups, the sleepers on each of a dozen streams in turn (torch's pool hands out streams that map onto different hardware queues).
    python tools/queue_probe.py          (GPU; GPU_MAX_HW_QUEUES=8 python tools/queue_probe.py for the runtime's other mapping)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import sgg_amd
from sgg_amd.rel_model_base import to_device_with_mirror
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
from sgg_amd.trainer import Trainer


def main():
    dev = 'cuda:0'
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'native', 'libspin.so'))
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(dev).eval()
    model.set_compute_dtype(torch.float16)
    b = list(synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=111))
    b[0] = [(im * 255).round().to(torch.uint8).permute(1, 2, 0).contiguous().to(dev) for im in b[0]]
    b[3], b[4], b[5] = b[3].to(dev), to_device_with_mirror(b[4], dev), to_device_with_mirror(b[5], dev)
    b = tuple(b)
    tr = Trainer(model, lr=1e-3, pipeline=True)
    for _ in range(10):
        tr.step(b)
    tr.flush()
    torch.cuda.synchronize()
    g = tr.graphs
    V = next(iter(g.V.values()))
    U = next(iter(g.U.values()))[0]
    sink = torch.zeros(4, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    streams = [torch.cuda.Stream() for _ in range(12)]
    hi = [torch.cuda.Stream(priority=-1) for _ in range(2)]

    def run(name, side, beside, main_stream=None, reps=12):
        ts = []
        ms = main_stream or torch.cuda.current_stream()
        for _ in range(reps):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(ms):
                e0.record()
                if beside is not None:
                    side.wait_stream(ms)
                    with torch.cuda.stream(side):
                        beside(side)
                V.replay()
                e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        print('%-78s VGG forward %.3f ms (min %.3f)' % (name, ts[len(ts) // 2], ts[0]), flush=True)
    print('GPU_MAX_HW_QUEUES =', os.environ.get('GPU_MAX_HW_QUEUES'))
    run('alone (default stream)', None, None)
    for blocks in (256, 16, 1):
        run('beside %3d sleeping workgroups (2.5 ms) on stream 0' % blocks, streams[0],
            lambda s, blocks=blocks: lib.spin(blocks, 250000, P(sink), ctypes.c_void_p(s.cuda_stream)))
    for k, s_ in enumerate(streams):
        run('beside 1 sleeping workgroup on stream %2d (handle %x)' % (k, s_.cuda_stream), s_,
            lambda s: lib.spin(1, 250000, P(sink), ctypes.c_void_p(s.cuda_stream)))
    run('beside 1 sleeping workgroup on a high-priority stream', hi[0], lambda s: lib.spin(1, 250000, P(sink), ctypes.c_void_p(s.cuda_stream)))
    # the forward itself on a pool stream instead of the default stream
    for k in (1, 2, 3):
        run('forward on pool stream %d, alone' % k, None, None, main_stream=streams[k])
        run('forward on pool stream %d beside 1 sleeping workgroup on stream 0' % k, streams[0],
            lambda s: lib.spin(1, 250000, P(sink), ctypes.c_void_p(s.cuda_stream)), main_stream=streams[k])
    lane = None
    from sgg_amd.imp import node_lane
    ln = node_lane(torch.device(dev))
    if ln is not None:
        lane = ln[0]
        run('beside 1 sleeping workgroup on the node lane (handle %x)' % lane.cuda_stream, lane,
            lambda s: lib.spin(1, 250000, P(sink), ctypes.c_void_p(s.cuda_stream)))
        run('beside the update (graph U) on the node lane', lane, lambda s: U.replay())
    for k in (0, 1, 2, 3, 5):
        run('beside the update (graph U) on stream %d' % k, streams[k], lambda s: U.replay())


if __name__ == '__main__':
    main()
