"""Plain launch vs the persistent stream-K form (csrc/tile_sched.h) of the 256x256 ping-pong kernel on the step's shapes:
us per launch (events around `reps` back-to-back launches) and TFLOP/s.   python tools/streamk_bench.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sgg_amd import ops

DEV = 'cuda:0'


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    g = torch.Generator().manual_seed(0)
    shapes = [('fc6 dW (pairs)', 4096, 25088, 3968, torch.bfloat16), ('fc7 fwd', 7936, 4096, 4096, torch.float16),
              ('fc6 fwd (pairs)', 3968, 4096, 25088, torch.float32), ('fc7 dX', 7936, 4096, 4096, torch.float16),
              ('300 tiles K=1024', 256 * 20, 256 * 15, 1024, torch.float16), ('496 tiles K=512', 7936, 4096, 512, torch.float16)]
    for name, M, N, K, odt in shapes:
        A = torch.randn(M, K, generator=g).to(DEV).half()
        W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV).half()
        out = torch.empty(M, N, dtype=odt, device=DEV)
        fn = lambda: ops.gemm(A, W, out=out, out_dtype=odt)     # noqa: E731
        tp, ts = [], []
        same = True
        for rnd in range(4):                                   # alternate: clocks and caches drift between back-to-back measurements
            ops.streamk_disable()
            tp.append(timed(fn, reps))
            ref = out.clone()
            prev = ops.streamk_mode(1)
            ops.streamk_enable()
            ts.append(timed(fn, reps))
            same = same and torch.equal(out, ref)
            ops.streamk_mode(prev)
        t_plain, t_sk = min(tp), min(ts)
        fl = 2.0 * M * N * K
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        print('%-18s %5d tiles (%.3f rounds)  plain %8.1f us %7.1f TF | stream-K %8.1f us %7.1f TF | %+5.1f %%  bits equal: %s'
              % (name, tiles, tiles / 256.0, t_plain, fl / t_plain / 1e6, t_sk, fl / t_sk / 1e6, 100 * (t_plain / t_sk - 1), same))
        if name.startswith('fc6 dW'):
            ops.streamk_disable()
            t_fw = timed(lambda: ops.gemm_full_waves(A, W, out_dtype=odt), reps)
            print('%-18s round 3 form (6 rounds + split-K tail): %8.1f us %7.1f TF' % ('', t_fw, fl / t_fw / 1e6))
        del A, W, out
    ops.streamk_disable()


if __name__ == '__main__':
    main()
