"""Where a stream-K workgroup's time goes (experiment build: make -C sgg_amd/csrc trace, then SGG_HIP_LIB=sgg_amd/libsgg_hip_trace.so):
s_memrealtime stamps (100 MHz) of workgroup 0 / wave 0 at the phases of every work item of one launch.   python tools/sk_trace.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sgg_amd import _lib, ops

DEV = 'cuda:0'


def main():
    lib = _lib.load()
    lib.sgg_sk_trace_buffer.argtypes = [ctypes.c_void_p]
    lib.sgg_sk_trace_buffer.restype = ctypes.c_int
    buf = torch.zeros(64 * 8, dtype=torch.int64, device=DEV)
    g = torch.Generator().manual_seed(0)
    for name, M, N, K, odt in (('fc6 dW', 4096, 25088, 3968, torch.bfloat16), ('fc7 fwd', 7936, 4096, 4096, torch.float16)):
        A = torch.randn(M, K, generator=g).to(DEV).half()
        W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV).half()
        out = torch.empty(M, N, dtype=odt, device=DEV)
        ops.streamk_mode(1)
        ops.streamk_enable()
        for _ in range(3):
            ops.gemm(A, W, out=out, out_dtype=odt)
        torch.cuda.synchronize()
        buf.zero_()
        assert lib.sgg_sk_trace_buffer(buf.data_ptr()) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm(A, W, out=out, out_dtype=odt)
        e1.record()
        torch.cuda.synchronize()
        lib.sgg_sk_trace_buffer(None)
        t = buf.view(64, 8).cpu().numpy()
        print('%s: launch %.1f us' % (name, e0.elapsed_time(e1) * 1e3))
        t0 = t[0, 0]
        for i in range(64):
            if t[i, 0] == 0:
                break
            st = [(int(x) - int(t0)) / 100.0 if x else None for x in t[i, :5]]
            k = (st[2] - st[1]) if st[2] is not None else None
            print('  item %d: start %8.2f  k-loop begins %8.2f  ends %8.2f (%.2f us)  staging starts %s  epilogue ends %s' %
                  (i, st[0], st[1], st[2], k, '%8.2f' % st[3] if st[3] is not None else '   -', '%8.2f' % st[4] if st[4] is not None else '   -'))
    ops.streamk_mode(0)


if __name__ == '__main__':
    main()
