"""which graphs / halves / column slices of the read stream differ from the short-lived kernel (debugging aid)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'oracle'))
from sgg_amd import ops  # noqa: E402

dev = 'cuda:0'
H = 128
dtype = torch.bfloat16
sizes = [int(x) for x in sys.argv[1:]] or [32, 5, 17, 2, 32, 9, 31, 3, 12]
im = torch.cat([torch.full((n,), b) for b, n in enumerate(sizes)]).to(dev)
rel, _ = ops.pair_index_eval(im)
N, E = len(im), len(rel)
per_graph = [n * (n - 1) for n in sizes]
g = torch.Generator().manual_seed(77)
v = torch.randn(N, H, generator=g).to(dtype).to(dev)
e = torch.randn(E, H, generator=g).to(dtype).to(dev)
nd, ed, gb = torch.randn(N, 4, generator=g).to(dev), torch.randn(E, 4, generator=g).to(dev), torch.randn(4, generator=g).to(dev)
csr = ops.edge_csr(rel, N, im, graphs=(len(sizes), max(sizes), max(per_graph)))
os.environ['SGG_IMP_STREAM'] = '0'
_, ref = ops.imp_sliced(v, e, csr, nd, ed, gb)
gates = torch.empty((E, 2), device=dev)
ops.imp_edge_in(v, csr, nd, ed, gb, E, gates_oi=gates)
got = ops.imp_ctx(e, csr, N, gates)
torch.cuda.synchronize()
d = (got.float() - ref.float()).abs()
n0 = 0
for b, n in enumerate(sizes):
    for half in (0, 1):
        blk = d[half, n0:n0 + n]
        print('graph %d (%d nodes) %s: max diff %.3f  per 64-column slice %s  bad nodes %s' % (
            b, n, ('out', 'in')[half], blk.max().item(), [round(blk[:, c:c + 64].max().item(), 2) for c in range(0, H, 64)],
            (blk.max(1).values > 0.2).nonzero().flatten().tolist()[:40]))
    n0 += n
