"""conv1_2 (Cin 64 -> Cout 64, fused 2x2 pool, 608x608, bf16): where do non-reproducible outputs sit?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from sgg_amd import ops
dt = torch.bfloat16
B, H, W, ci, co = int(os.environ.get('B', '4')), 608, 608, 64, 64
pool = os.environ.get('POOL', '1') == '1'
g = torch.Generator().manual_seed(0)
x = torch.zeros((B, H + 2, W + 2, ci), dtype=dt, device='cuda:0')
x[:, 1:-1, 1:-1] = torch.randn(B, H, W, ci, generator=g).to('cuda:0').to(dt)
w = (torch.randn(co, 3, 3, ci, generator=g) / 24).to('cuda:0').to(dt)
bias = torch.randn(co, generator=g).to('cuda:0')
shape = (B, H // 2 + 2, W // 2 + 2, co) if pool else (B, H + 2, W + 2, co)
ref = torch.zeros(shape, dtype=dt, device='cuda:0')
ops.conv3x3_relu(x, w, bias, ref, 1, pool=pool)
torch.cuda.synchronize()
# trusted value: torch conv in fp32 on the GPU
xt = x[:, 1:-1, 1:-1].permute(0, 3, 1, 2).float()
yt = torch.nn.functional.conv2d(xt, w.permute(0, 3, 1, 2).float(), bias, padding=1).relu()
if pool:
    yt = torch.nn.functional.max_pool2d(yt, 2)
yt = yt.permute(0, 2, 3, 1)
print('first run vs torch fp32: max |diff| %.4f' % float((ref[:, 1:-1, 1:-1].float() - yt).abs().max()))
bad = 0
for r in range(int(os.environ.get('REPS', '200'))):
    y = torch.zeros(shape, dtype=dt, device='cuda:0')
    ops.conv3x3_relu(x, w, bias, y, 1, pool=pool)
    torch.cuda.synchronize()
    if not torch.equal(y, ref):
        bad += 1
        d = (y.float() - ref.float()).abs()
        idx = (d > 0).nonzero()
        e_new = float((y[:, 1:-1, 1:-1].float() - yt).abs().max())
        print('run %d: %d elements differ, max %.3f; this run vs torch %.4f; b %s y %d..%d x %d..%d c %d..%d' % (
            r, idx.shape[0], float(d.max()), e_new, sorted(set(idx[:, 0].tolist())), int(idx[:, 1].min()), int(idx[:, 1].max()),
            int(idx[:, 2].min()), int(idx[:, 2].max()), int(idx[:, 3].min()), int(idx[:, 3].max())), flush=True)
print('differing runs: %d' % bad)
