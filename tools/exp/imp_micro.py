"""IMP sliced-step kernels without the model: random states on complete 32-node graphs, hipGraph-replay timing of each form;
with a library built with -DSGG_DMA_TICKS (SGG_HIP_LIB=...), prints the per-chunk phase stamps of the read stream.
    python tools/exp/imp_micro.py [B ...]      env: IMP_FORMS=0,s,A,B,par,1  IMP_DTYPE=bf16|f32  IMP_EAGER=1"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sgg_amd import _lib, ops  # noqa: E402

dev = 'cuda:0'
dtype = torch.float32 if os.environ.get('IMP_DTYPE') == 'f32' else torch.bfloat16
s = 2 if dtype == torch.bfloat16 else 4
H, n = int(os.environ.get('IMP_H', '512')), 32
forms = os.environ.get('IMP_FORMS', '0,s,A,B').split(',')
Bs = [int(b) for b in sys.argv[1:]] or [8, 128]
for B in Bs:
    N, E = n * B, n * (n - 1) * B
    im = torch.arange(B, device=dev).repeat_interleave(n)
    rel, _ = ops.pair_index_eval(im)
    rel = rel[:E]
    csr = ops.edge_csr(rel, N, im, graphs=(B, n, n * (n - 1)))
    g = torch.Generator().manual_seed(1)
    v = torch.randn(N, H, generator=g).to(dev).to(dtype)
    e = torch.randn(E, H, generator=g).to(dev).to(dtype)
    nd, ed, gb = torch.randn(N, 4, generator=g).to(dev), torch.randn(E, 4, generator=g).to(dev), torch.randn(4, generator=g).to(dev)
    e_in, ctx2 = torch.empty_like(e), torch.empty((2, N, H), dtype=dtype, device=dev)
    algo = (2.0 * (E + N) * H) * s + 8.0 * E
    row = []
    side = torch.cuda.Stream()
    ev0, ev1 = torch.cuda.Event(), torch.cuda.Event()

    def par():                      # the split step on two streams: write stream here, read stream on the side
        cur = torch.cuda.current_stream()
        ev0.record(cur)
        side.wait_event(ev0)
        with torch.cuda.stream(side):
            ops.imp_ctx(e, csr, N, gates, ctx2=ctx2)
            ev1.record(side)
        ops.imp_edge_in(v, csr, nd, ed, gb, E, e_in=e_in)
        cur.wait_event(ev1)
    gates = torch.empty((E, 2), device=dev)
    ops.imp_edge_in(v, csr, nd, ed, gb, E, e_in=e_in, gates_oi=gates)
    for form in forms:
        os.environ['SGG_IMP_STREAM'] = form if form not in ('A', 'B', 'par', 'm') else '0'
        launch = {'A': lambda: ops.imp_edge_in(v, csr, nd, ed, gb, E, e_in=e_in, gates_oi=gates), 'B': lambda: ops.imp_ctx(e, csr, N, gates, ctx2=ctx2),
                  's': lambda: ops.imp_split(v, e, csr, nd, ed, gb, e_in=e_in, ctx2=ctx2, gates_oi=gates),
                  'm': lambda: ops.imp_step(v, e, csr, nd, ed, gb, e_in=e_in, ctx2=ctx2),
                  'par': par}.get(form, lambda: ops.imp_sliced(v, e, csr, nd, ed, gb, e_in, ctx2))
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(50):
                launch()
        graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        row.append('%s %7.2f us %5.0f GB/s %.3f' % (form, ms * 1e3, algo / ms / 1e6, algo / ms / 1e6 / 8000))
    print('B=%3d' % B, ' | '.join(row), flush=True)
    if os.environ.get('IMP_EAGER'):                     # no graph: do the two streams of 'par' overlap on the device?
        for name, fn in (('s', lambda: ops.imp_split(v, e, csr, nd, ed, gb, e_in=e_in, ctx2=ctx2, gates_oi=gates)), ('par', par)):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                fn()
            e1.record()
            torch.cuda.synchronize()
            print('   eager %s: %.2f us per step' % (name, e0.elapsed_time(e1) * 5))
    lib = _lib.load()
    if hasattr(lib, 'sgg_dbg_dma_ticks') and ('B' in forms or 'm' in forms):
        if 'm' in forms:
            ops.imp_step(v, e, csr, nd, ed, gb, e_in=e_in, ctx2=ctx2)
        else:
            ops.imp_ctx(e, csr, N, gates, ctx2=ctx2)
        torch.cuda.synchronize()
        buf = np.zeros(8 * 16 * 8, dtype=np.int64)
        lib.sgg_dbg_dma_ticks.argtypes = [ctypes.c_void_p]
        assert lib.sgg_dbg_dma_ticks(buf.ctypes.data) == 0
        t = buf.reshape(8, 16, 8)
        for blk in (0, 3):
            print(' block %d (wave 0; clock64 ticks per CHUNK of the read stream)' % blk)
            prev_end = None
            for u in range(16):
                r = t[blk, u]
                if r[4] == 0:
                    break
                gap = (r[0] - prev_end) if prev_end is not None else 0
                print('   chunk %2d: gap %6d | wait+barrier(s) %6d  dma-issue %5d  out-lists %6d  in-lists %6d | unit end: wait E %6d  sums+stores %6d' %
                      (u, gap, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], (r[5] - r[4]) if r[5] else 0, (r[6] - r[5]) if r[6] else 0))
                prev_end = r[6] if r[6] else r[4]
    os.environ.pop('SGG_IMP_STREAM', None)
