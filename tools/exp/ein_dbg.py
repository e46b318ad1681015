import os, sys, torch
sys.path.insert(0, '/root/repo')
from sgg_amd import ops
dev='cuda:0'; H=128; dtype=torch.bfloat16
sizes=[32,5,17,2,32,9,31,3,12]
im=torch.cat([torch.full((n,),b) for b,n in enumerate(sizes)]).to(dev)
rel,cnt=ops.pair_index_eval(im); rel=rel[:int(cnt.item())]; N,E=len(im),len(rel)
g=torch.Generator().manual_seed(77)
v=torch.randn(N,H,generator=g).to(dtype).to(dev); e=torch.randn(E,H,generator=g).to(dtype).to(dev)
nd,ed,gb=torch.randn(N,4,generator=g).to(dev),torch.randn(E,4,generator=g).to(dev),torch.randn(4,generator=g).to(dev)
csr=ops.edge_csr(rel,N,im,graphs=(len(sizes),max(sizes),max(n*(n-1) for n in sizes)))
os.environ['SGG_IMP_STREAM']='0'
ref_ein,ref_ctx=ops.imp_sliced(v,e,csr,nd,ed,gb)
ein,ctx=ops.imp_step(v,e,csr,nd,ed,gb)
torch.cuda.synchronize()
d=(ein.float()-ref_ein.float()).abs()
print('e_in max diff', d.max().item(), 'mismatched', int((d>0).sum()), 'of', d.numel(), 'nan', int(torch.isnan(ein.float()).sum()))
bad=(d>0).nonzero()[:8]
for r,c in bad.tolist(): print(r,c, ein[r,c].item(), ref_ein[r,c].item())
print('ctx max diff', (ctx.float()-ref_ctx.float()).abs().max().item())
nanrow = torch.isnan(ein.float()).any(1)
e0 = 0
for b, n in enumerate(sizes):
    ne = n * (n - 1)
    blk = torch.isnan(ein[e0:e0 + ne].float())
    print('graph %d (%d nodes, %d edges): NaN rows %d, NaN cols per 64-slice %s; ctx NaN %d' % (
        b, n, ne, int(blk.any(1).sum()), [int(blk[:, c:c + 64].any()) for c in range(0, H, 64)],
        int(torch.isnan(ctx[:, sum(sizes[:b]):sum(sizes[:b + 1])].float()).sum())))
    e0 += ne
print('row 0 of e_in:', ein[0, :16].float().tolist())
print('ref row 0     :', ref_ein[0, :16].float().tolist())
print('NaN per column (first 16):', torch.isnan(ein.float()).float().mean(0)[:16].tolist())
v1 = torch.ones_like(v)
ein1, _ = ops.imp_step(v1, e, csr, nd, ed, gb)
ref1, _ = ops.imp_sliced(v1, e, csr, nd, ed, gb)
print('v=1: row 0', ein1[0, :8].float().tolist(), 'ref', ref1[0, :8].float().tolist(), 'NaN frac', torch.isnan(ein1.float()).float().mean().item())
