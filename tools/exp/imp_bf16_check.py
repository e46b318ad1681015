import numpy as np, torch, sys
sys.path.insert(0, '.')
from oracle import sgg_oracle as O
from sgg_amd import ops
cu = lambda t: (torch.from_numpy(np.ascontiguousarray(t)) if isinstance(t, np.ndarray) else t.contiguous()).to("cuda:0")
H = 512
g = torch.Generator().manual_seed(3)
im = np.repeat(np.arange(2), 32).astype(np.int64)
rel = O.get_rel_inds_eval(im)
N, E = len(im), len(rel)
gw, gb = torch.randn(4, 2 * H, generator=g) / (H ** 0.5), torch.randn(4, generator=g)
for scale in (1.0, 0.3):
    v, e = (torch.randn(N, H, generator=g) * scale).bfloat16(), (torch.randn(E, H, generator=g) * scale).bfloat16()
    s, o = torch.from_numpy(rel[:, 1]), torch.from_numpy(rel[:, 2])
    vf, ef = v.float(), e.float()
    for wdt in ('f32w', 'bf16w'):
        w = gw if wdt == 'f32w' else gw.bfloat16().float()
        gt = [torch.sigmoid(torch.cat((a, ef), 1) @ w[k] + gb[k]) for k, a in enumerate((vf[s], vf[o], vf[s], vf[o]))]
        exp = gt[0][:, None] * vf[s] + gt[1][:, None] * vf[o]
        csr = ops.edge_csr(cu(rel), N, cu(im))
        e_in, ctx2 = ops.imp_fused(cu(v), cu(e), cu(rel), csr, cu(gw).bfloat16(), cu(gb))
        d = (e_in.float().cpu() - exp).abs()
        print(scale, wdt, 'fused bf16 vs expectation: max', float(d.max()), 'mean', float(d.mean()))
