"""Do the write stream (imp_edge_in) and the read stream (imp_ctx) of the split IMP step overlap when they sit on two HIP streams with
no dependency between them?  20 launches of each: back to back on one stream vs one stream each."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sgg_amd import ops  # noqa: E402

dev, dtype, H, n = 'cuda:0', torch.bfloat16, 512, 32
for B in [int(b) for b in sys.argv[1:]] or [128]:
    N, E = n * B, n * (n - 1) * B
    im = torch.arange(B, device=dev).repeat_interleave(n)
    rel, _ = ops.pair_index_eval(im)
    csr = ops.edge_csr(rel[:E], N, im, graphs=(B, n, n * (n - 1)))
    g = torch.Generator().manual_seed(1)
    v = torch.randn(N, H, generator=g).to(dev).to(dtype)
    e = torch.randn(E, H, generator=g).to(dev).to(dtype)
    nd, ed, gb = torch.randn(N, 4, generator=g).to(dev), torch.randn(E, 4, generator=g).to(dev), torch.randn(4, generator=g).to(dev)
    e_in, ctx2, gates = torch.empty_like(e), torch.empty((2, N, H), dtype=dtype, device=dev), torch.empty((E, 2), device=dev)
    A = lambda: ops.imp_edge_in(v, csr, nd, ed, gb, E, e_in=e_in, gates_oi=gates)      # noqa: E731
    Bk = lambda: ops.imp_ctx(e, csr, N, gates, ctx2=ctx2)                                # noqa: E731
    A(); Bk()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def timed(fa, fb, two):
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0.record()
        s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s1):
            for _ in range(20):
                fa()
                if not two:
                    fb()
        if two:
            with torch.cuda.stream(s2):
                for _ in range(20):
                    fb()
        torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
        t1.record()
        torch.cuda.synchronize()
        return t0.elapsed_time(t1) * 1e3 / 20
    for _ in range(2):
        one, two = timed(A, Bk, False), timed(A, Bk, True)
    onlyA, onlyB = timed(A, lambda: None, False), timed(Bk, lambda: None, False)
    print('B=%d: A alone %.1f us, B alone %.1f us, A then B on one stream %.1f us, A and B on two streams %.1f us per pair' % (B, onlyA, onlyB, one, two))
