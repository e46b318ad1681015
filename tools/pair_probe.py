"""What slows the VGG-16 forward down beside the optimiser's update (2.5 ms for the pair against 1.75 + 1.7 ms alone)?  The forward (graph V
of a replayed step) is timed alone and beside: workgroups that only occupy wave slots (one per CU, sleeping), workgroups that stream
memory at several grid sizes, and the update itself (graph U).     python tools/pair_probe.py"""
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
    side = torch.cuda.Stream()
    sink = torch.zeros(4, device=dev)
    n4 = (1 << 30) // 16
    src, dst = torch.empty(1 << 30, dtype=torch.uint8, device=dev), torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())

    def run(name, beside, reps=20):
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            e0, e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if beside is not None:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    beside(side)
                    e2.record(side)
            V.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append((e0.elapsed_time(e1), e0.elapsed_time(e2) if beside is not None else 0.0))
        ts.sort()
        v, o = ts[len(ts) // 2]
        print('%-64s VGG forward %.3f ms   the other %.3f ms' % (name, v, o), flush=True)
    # is it the kernel BOUNDARIES of the forward (14 launches) or co-residency itself?  one long MFMA kernel instead of the forward:
    from sgg_amd import ops
    A = torch.randn(15872, 25088, device=dev).half()
    Wb = torch.randn(4096, 25088, device=dev).half()
    outb = torch.empty(15872, 4096, device=dev, dtype=torch.float16)

    def run_gemm(name, beside, reps=10):
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if beside is not None:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    beside(side)
            ops.gemm(A, Wb, out=outb)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        print('%-64s one GEMM launch (3.3 TFLOP) %.3f ms' % (name, ts[len(ts) // 2]), flush=True)
    run_gemm('alone', None)
    run_gemm('beside 256 sleeping workgroups (3 ms)', lambda s: lib.spin(256, 300000, P(sink), ctypes.c_void_p(s.cuda_stream)))
    run_gemm('beside a 4 GiB copy stream on 256 workgroups x3', lambda s: [lib.stream_copy(256, P(src), P(dst), n4, 2, ctypes.c_void_p(s.cuda_stream)) for _ in range(3)])
    del A, Wb, outb
    run('alone', None)
    for blocks in (256, 512, 1024):
        run('beside %4d sleeping workgroups (2.2 ms, no memory traffic)' % blocks, lambda s, blocks=blocks: lib.spin(blocks, 220000, P(sink), ctypes.c_void_p(s.cuda_stream)))
    for blocks in (32, 64, 128, 256, 1024):
        run('beside a 4 GiB copy stream on %4d workgroups' % blocks,
            lambda s, blocks=blocks: lib.stream_copy(blocks, P(src), P(dst), n4, 2, ctypes.c_void_p(s.cuda_stream)))
    run('beside the update (graph U)', lambda s: U.replay())
    # the update's own access pattern as a bare streaming kernel: 250 M parameters, p / m f32, g / shadow f16 (5 GB per pass)
    npar = 250 * (1 << 20)
    pp, mm = torch.zeros(npar, device=dev), torch.zeros(npar, device=dev)
    gg, ss = torch.zeros(npar, device=dev, dtype=torch.float16), torch.empty(npar, device=dev, dtype=torch.float16)
    cnt = torch.zeros(4, dtype=torch.int32, device=dev)
    for blocks, mode in ((256, 0), (256, 1), (512, 0), (1024, 0)):
        def sgd(s, blocks=blocks, mode=mode):
            lib.sgd_stream(blocks, P(pp), P(mm), P(gg), P(ss), ctypes.c_long(npar), mode, P(cnt), ctypes.c_void_p(s.cuda_stream))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sgd(torch.cuda.current_stream())
        e0.record()
        sgd(torch.cuda.current_stream())
        e1.record()
        torch.cuda.synchronize()
        print('bare update stream alone, %d workgroups, %s chunks: %.3f ms for 5.24 GB (%.2f TB/s)' % (
            blocks, 'static' if mode == 0 else 'fetched', e0.elapsed_time(e1), 5.243 / e0.elapsed_time(e1)), flush=True)
        run('beside the bare update stream (%d workgroups, %s)' % (blocks, 'static' if mode == 0 else 'fetched'), sgd)
    del pp, mm, gg, ss
    # a streaming kernel that claims a few CUs by their hardware id (tools/native/spin.hip): what does it reach, what does it cost the forward?
    state = torch.zeros(4096, dtype=torch.int32, device=dev)
    for cus, workers in ((4, 4), (4, 8), (6, 4), (8, 4), (8, 2), (32, 1)):
        def claim(s, cus=cus, workers=workers):
            lib.claim_stream_copy(8192, P(src), P(dst), n4, 2, P(state), cus, workers, ctypes.c_void_p(s.cuda_stream))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        claim(torch.cuda.current_stream())
        e1.record()
        torch.cuda.synchronize()
        st = state.tolist()
        print('claiming copy alone: %d CUs per XCD x %d workers: %.3f ms for 4.3 GB (%.2f TB/s); CUs claimed per XCD %s, workers %d' % (
            cus, workers, e0.elapsed_time(e1), 4.295 / e0.elapsed_time(e1), st[4032:4040], st[4041]), flush=True)
        run('beside the claiming copy (%d CUs per XCD x %d workers)' % (cus, workers), claim)


if __name__ == '__main__':
    main()
