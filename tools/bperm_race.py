"""Is the ds_bpermute butterfly (__shfl_xor) reliable while another stream's kernels share the chip?  (Round 5: the f32 gate kernel's dot
products were wrong in ~15 % of its launches beside the x3 mode's f16 GEMM -- tools/gate_race.py; the DPP form that replaced the butterfly
is not.)  A kernel that does nothing but butterflies of known values (tools/native/bperm_check.hip) runs on the main stream while a second
stream runs one kind of work; reports the lanes that saw a wrong total.
    python tools/bperm_race.py      (GPU, ~20 s)
"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgg_amd import ops  # noqa: E402

DEV = 'cuda:0'
REPS = int(os.environ.get('REPS', '300'))


def main():
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'native', 'libbperm_check.so'))
    torch.manual_seed(0)
    N, H = 256, 512
    seed = torch.tensor([1.0, 2.0, 3.0, 4.0], device=DEV)
    errors = torch.zeros(2, dtype=torch.int32, device=DEV)
    sink = torch.zeros(1, device=DEV)
    obj = torch.randn(N, H, device=DEV)
    W = (torch.randn(3 * H, H, device=DEV) * 0.04).contiguous()
    obj48, W48 = torch.randn(N, 3 * H, device=DEV).half(), (torch.randn(3 * H, 3 * H, device=DEV) * 0.02).half()
    out48 = torch.empty(N, 3 * H, device=DEV)
    big = torch.randn(4096, 4096, device=DEV)
    side = torch.cuda.Stream(device=DEV)

    def x3_gemm():
        prev = ops.set_split3(True)
        ops.gemm(obj, W, None, out_dtype=torch.float32)
        ops.set_split3(prev)

    works = (('nothing', None), ('x3 gemm (split3 x2 + f16 gemm)', x3_gemm), ('split3 only', lambda: (ops.split3(obj), ops.split3(W, weights=True))),
             ('f16 gemm K=1536 -> f32, no allocation', lambda: ops.gemm(obj48, W48, None, out_dtype=torch.float32, out=out48)),
             ('f32 gemm', lambda: ops.gemm(obj, W, None, out_dtype=torch.float32)), ('torch.mm 4096', lambda: torch.mm(big, big)),
             ('x3 gemm again', x3_gemm))
    report = {}
    for width in (64, 16):
        for name, work in works:
            errors.zero_()
            for r in range(REPS):
                if work is not None:
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        for _ in range(3):
                            work()
                rc = lib.bperm_check(1984, 6, width, ctypes.c_void_p(seed.data_ptr()), ctypes.c_void_p(errors.data_ptr()), ctypes.c_void_p(sink.data_ptr()),
                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                assert rc == 0
                torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            e = errors.tolist()
            report['%d: %s' % (width, name)] = dict(launches=REPS, lanes_with_a_wrong_total=e[0], which_of_the_four_values='%08x' % (e[1] & 0xffffffff))
            print('width %2d  %-42s lanes with a wrong total: %6d of %d x %d   (byte k of %08x: value k wrong in some lane)' % (
                width, name, e[0], REPS, 1984 * 256, e[1] & 0xffffffff))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'bperm_race.json'), 'w') as f:
        json.dump(report, f, indent=1)


if __name__ == '__main__':
    main()
