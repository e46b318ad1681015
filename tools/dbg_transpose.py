import torch, sys
sys.path.insert(0, '.')
from sgg_amd import ops, _lib
g = torch.Generator().manual_seed(2)
R, C = 128, 128
xo = torch.randn(R, C, generator=g)
for src, dt in ((xo, torch.bfloat16), (xo.bfloat16(), torch.bfloat16), (xo, torch.float32)):
    x = src.cuda()
    out = torch.zeros((C, R), dtype=dt, device='cuda')
    rc = _lib.call('sgg_transpose', x.data_ptr(), x.stride(0), out.data_ptr(), R, R, C, None, 0, 1, None, ops.dt(x), ops.dt(out),
              torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    exp = src.float().t().to(dt)
    d = (out.cpu().float() - exp.float())
    print(src.dtype, dt, 'rc', rc, 'nan', int(torch.isnan(out).sum()), 'maxdiff', float(d.nan_to_num(9).abs().max()), out[0, :4].tolist(), exp[0, :4].tolist())
