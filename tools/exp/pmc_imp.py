"""One form of the sliced IMP step, a few launches at B images (for rocprofv3 --pmc passes).  usage: pmc_imp.py FORM B"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sgg_amd import ops  # noqa: E402

form, B = sys.argv[1], int(sys.argv[2])
os.environ['SGG_IMP_STREAM'] = form
dev, dtype, H, n = 'cuda:0', torch.bfloat16, 512, 32
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
for _ in range(5):
    ops.imp_sliced(v, e, csr, nd, ed, gb, e_in, ctx2)
torch.cuda.synchronize()
print('done')
