"""Which VGG layer is not bit-reproducible run to run?  Every conv of the bf16 detector is run REPS times on one fixed input."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sgg_amd
from sgg_amd import ops
from sgg_amd.detector import VGG16_CFG
from sgg_amd.synthetic import SyntheticData, init_weights
REPS = int(os.environ.get('REPS', '40'))
dt = torch.bfloat16 if os.environ.get('DT', 'bf16') == 'bf16' else torch.float32
B = int(os.environ.get('B', '4'))
model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to('cuda:0').eval()
det = model.detector
ws = det.prepared(dt)
g = torch.Generator().manual_seed(0)
H = W = 608
x = torch.zeros((B, H + 2, W + 2, 4), dtype=torch.float32, device='cuda:0')
x[:, 1:-1, 1:-1, :3] = torch.randn(B, H, W, 3, generator=g).to('cuda:0')
ci_layer, li, cfg = 0, 0, list(VGG16_CFG)
while li < len(cfg):
    v = cfg[li]
    if v == 'M':
        y = torch.zeros((B, H // 2 + 2, W // 2 + 2, x.shape[3]), dtype=dt, device='cuda:0')
        ops.maxpool2x2(x, y, 1)
        H, W = H // 2, W // 2
        name = 'pool'
        run = None
    else:
        w, bias, ci, co = ws[ci_layer]
        last = ci_layer == len(ws) - 1
        fuse = (ci_layer > 0 and li + 1 < len(cfg) and cfg[li + 1] == 'M' and ops.conv_pool_fusable(H, W, co))
        if fuse:
            shape = (B, H // 2 + 2, W // 2 + 2, co)
            run = lambda y, x=x, w=w, bias=bias, ci=ci, co=co: ops.conv3x3_relu(x, w.view(co, 3, 3, ci), bias, y, 1, pool=True)
            H, W = H // 2, W // 2
            li += 1
        elif ci_layer == 0:
            shape = (B, H + 2, W + 2, co)
            run = lambda y, x=x, w=w, bias=bias: ops.conv1_1(x, w, bias, y)
        else:
            shape = (B, H, W, co) if last else (B, H + 2, W + 2, co)
            run = lambda y, x=x, w=w, bias=bias, ci=ci, co=co, last=last: ops.conv3x3_relu(x, w.view(co, 3, 3, ci), bias, y, 0 if last else 1)
        name = 'conv%d %s Cin %d Cout %d out %s' % (ci_layer, 'fused-pool' if fuse else '', ci, co, shape[1:3])
        y = torch.zeros(shape, dtype=dt, device='cuda:0')
        run(y)
        torch.cuda.synchronize()
        bad = 0
        worst = 0.0
        for r in range(REPS):
            y2 = torch.zeros(shape, dtype=dt, device='cuda:0')
            run(y2)
            torch.cuda.synchronize()
            if not torch.equal(y, y2):
                bad += 1
                worst = max(worst, float((y.float() - y2.float()).abs().max()))
        print('%-60s runs differing from the first: %d / %d   max |diff| %.4f' % (name, bad, REPS, worst), flush=True)
        ci_layer += 1
    x = y
    li += 1
