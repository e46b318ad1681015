import os, sys, torch, numpy as np
sys.path.insert(0, '.')
import sgg_amd
from sgg_amd import ops
dev = 'cuda:0'
n, H, dt = 32, 512, torch.bfloat16
for B in (8, 128):
    N, E = n * B, n * (n - 1) * B
    im = torch.arange(B, device=dev).repeat_interleave(n)
    rel, _ = ops.pair_index_eval(im); rel = rel[:E]
    csr = ops.edge_csr(rel, N, im, graphs=(B, n, n * (n - 1)))
    v = torch.randn(N, H, device=dev).to(dt); e = torch.randn(E, H, device=dev).to(dt)
    gw = torch.randn(4, 2 * H, device=dev) / 30; gb = torch.randn(4, device=dev)
    nd, ed = (v.float() @ gw[:, :H].t()).contiguous(), (e.float() @ gw[:, H:].t()).contiguous()
    e_in, ctx2 = torch.empty_like(e), torch.empty((2, N, H), dtype=dt, device=dev)
    for _ in range(3):
        ops.imp_sliced(v, e, csr, nd, ed, gb, e_in, ctx2)
    torch.cuda.synchronize()
    raw = ctx2.view(torch.int64).cpu().numpy().reshape(-1)[:4 * 16 * 8].reshape(4, 16, 8)
    for blk in range(2):
        t = raw[blk]                       # [wave, tick]
        base = t[:, 0].min()
        print('B=%d block %d: ticks rel. to first wave start (min..max over waves)' % (B, blk))
        for i, name in enumerate(['start', 'lvl2 loads issued', 'parked', 'after barrier 1', 'phase 1 done', 'after barrier 2', 'phase 2 done', 'stores drained']):
            print('   %-18s %7d .. %7d' % (name, (t[:, i] - base).min(), (t[:, i] - base).max()))
