"""tools/x3_repro.py stress found the x3 mode's two-stream forward non-deterministic (5 - 10 % of the runs), first at the gate dot products
of the edge GRU's first call: ONE row's first dot product off by ~0.1, the state row itself right.  This isolates it: the gate kernel on
the main stream, repeated, while a second stream runs one kind of work; every result compared bit for bit with the first.
    python tools/gate_race.py        (GPU, ~20 s)
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgg_amd import ops  # noqa: E402

DEV = 'cuda:0'
REPS = int(os.environ.get('REPS', '300'))


def main():
    torch.manual_seed(0)
    E, N, H = 7936, 256, 512
    gi = torch.randn(E, 3 * H, device=DEV)
    b_hh = torch.randn(3 * H, device=DEV) * 0.1
    gate_w = (torch.randn(4, 2 * H, device=DEV) * 0.05).contiguous()
    we = gate_w[:, H:]
    obj = torch.randn(N, H, device=DEV)
    W = (torch.randn(3 * H, H, device=DEV) * 0.04).contiguous()
    big = torch.randn(4096, 4096, device=DEV)
    side = torch.cuda.Stream(device=DEV)
    ref_h, ref_d = ops.gru_gate(gi, None, b_hh, None, torch.float32, dot_w=we)
    torch.cuda.synchronize()
    truth = ref_h @ we.t()
    print('dots vs torch (f32 matmul): max abs %.3e' % float((ref_d - truth).abs().max()))

    def side_x3_gemm():
        prev = ops.set_split3(True)
        ops.gemm(obj, W, None, out_dtype=torch.float32)
        ops.set_split3(prev)

    def side_split_only():
        ops.split3(obj)
        ops.split3(W, weights=True)

    def side_f16_gemm():
        ops.gemm(obj16, W16, None, out_dtype=torch.float32)

    def side_f32_gemm():
        ops.gemm(obj, W, None, out_dtype=torch.float32)

    def side_node_gate():
        ops.gru_gate(gin, None, b_hh, None, torch.float32, dot_w=gate_w[:, :H])

    def side_torch_mm():
        torch.mm(big, big)

    def side_torch_elementwise():
        big.mul_(1.0)

    obj16, W16 = obj.half(), W.half()
    obj48, W48 = torch.randn(N, 3 * H, device=DEV).half(), (torch.randn(3 * H, 3 * H, device=DEV) * 0.02).half()
    out48 = torch.empty(N, 3 * H, device=DEV)

    def side_f16_gemm_k1536():
        ops.gemm(obj48, W48, None, out_dtype=torch.float32, out=out48)

    def side_f16_gemm_k1536_f16out():
        ops.gemm(obj48, W48, None)
    gin = torch.randn(N, 3 * H, device=DEV)
    report = {}
    for name, work in (('nothing', None), ('x3 gemm (split3 x2 + f16 gemm)', side_x3_gemm), ('split3 only', side_split_only),
                       ('f16 gemm only', side_f16_gemm), ('f32 gemm only', side_f32_gemm), ('node gate kernel', side_node_gate),
                       ('torch.mm 4096', side_torch_mm), ('torch elementwise', side_torch_elementwise),
                       ('f16 gemm K=1536 -> f32 (no allocation)', side_f16_gemm_k1536), ('f16 gemm K=1536 -> f16', side_f16_gemm_k1536_f16out),
                       ('x3 gemm again', side_x3_gemm)):
        bad, cols, h_bad = 0, {}, 0
        for r in range(REPS):
            if work is not None:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(3):
                        work()
            h, d = ops.gru_gate(gi, None, b_hh, None, torch.float32, dot_w=we)
            if not torch.equal(d, ref_d):
                bad += 1
                diff = (d != ref_d)
                for c in torch.nonzero(diff.any(0)).view(-1).tolist():
                    cols[c] = cols.get(c, 0) + 1
            if not torch.equal(h, ref_h):
                h_bad += 1
            torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        report[name] = dict(reps=REPS, dots_differ=bad, state_differs=h_bad, columns=cols)
        print('%-40s dots differ in %3d of %d launches (state rows: %d)  columns %s' % (name, bad, REPS, h_bad, cols))
    # the round-4 form of the epilogue (ds_bpermute butterfly), checking itself: tools/native/gate_check.hip
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'native', 'libgate_check.so')
    if os.path.exists(path):
        import ctypes
        lib = ctypes.CDLL(path)
        counts = torch.zeros(8, dtype=torch.int32, device=DEV)
        h2, d2 = torch.empty_like(ref_h), torch.empty_like(ref_d)
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        for name, work in (('nothing', None), ('x3 gemm (split3 x2 + f16 gemm)', side_x3_gemm), ('split3 only', side_split_only),
                           ('f16 gemm K=1536 -> f32 (no allocation)', side_f16_gemm_k1536), ('torch.mm 4096', side_torch_mm)):
            counts.zero_()
            first, differ = None, 0
            for r in range(REPS):
                if work is not None:
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        for _ in range(3):
                            work()
                rc = lib.gate_check(P(gi), P(b_hh), P(h2), E, H, ctypes.c_void_p(we.data_ptr()), we.stride(0), P(d2), P(counts),
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                assert rc == 0
                if first is None:
                    first = d2.clone()
                elif not torch.equal(first, d2):
                    differ += 1
                torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            c = counts.tolist()
            report['self-check: ' + name] = dict(launches=REPS, dots_differ_from_first_launch=differ, lanes_weight_reload_differs=c[0],
                                                 rows_butterfly_vs_dpp=c[1], rows_butterfly_lane0_vs_lane63=c[2], per_gate=c[3:7])
            print('self-check  %-40s dots differ %3d / %d   weight reload differs: %d lanes   butterfly != DPP: %d rows %s   lane 0 != lane 63: %d' % (
                name, differ, REPS, c[0], c[1], c[3:7], c[2]))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'gate_race.json'), 'w') as f:
        json.dump(report, f, indent=1)


if __name__ == '__main__':
    main()
