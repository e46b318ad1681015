"""Random shapes through sgg_conv3x3_relu (whatever kernel the library routes them to: conv_pp forms and tile widths, the patch kernel, the
implicit GEMMs) and sgg_conv1_block against torch's fp32 convolution of the same 16-bit operands.  `python tools/fuzz_conv.py [cases] [seed]`"""
import os
import random
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgg_amd import ops  # noqa: E402

dev = 'cuda:0'
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(cases):
    dt = rng.choice([torch.float16, torch.bfloat16])
    B = rng.randint(1, 3)
    pool = rng.random() < 0.4
    H, W = rng.randint(20, 200), rng.randint(20, 200)
    if pool:
        H, W = H // 2 * 2, W // 2 * 2
    if rng.random() < 0.15:                                   # the fused first block
        g = torch.Generator().manual_seed(case)
        img = torch.zeros(B, H + 2, W + 2, 4)
        img[:, 1:-1, 1:-1, :3] = torch.randn(B, H, W, 3, generator=g)
        w1, b1 = torch.randn(64, 27, generator=g) / 5, torch.randn(64, generator=g) * 0.1
        w2, b2 = (torch.randn(64, 3, 3, 64, generator=g) / 24).to(dt), torch.randn(64, generator=g) * 0.1
        y1 = F.relu(F.conv2d(img[:, 1:-1, 1:-1, :3].permute(0, 3, 1, 2), w1.view(64, 3, 3, 3).permute(0, 3, 1, 2), b1, padding=1)).to(dt).float()
        ref = F.relu(F.conv2d(y1, w2.float().permute(0, 3, 1, 2), b2, padding=1))
        if pool:
            ref = F.max_pool2d(ref, 2)
        out = torch.zeros(B, ref.shape[2] + 2, ref.shape[3] + 2, 64, device=dev, dtype=dt)
        ops.conv1_block(img.to(dev), ops.conv1_pack_weights(w1.to(dev), dt), b1.to(dev), w2.to(dev), b2.to(dev), out, 1, pool=pool)
        what = 'conv1_block'
        Cin = Cout = 64
    else:
        Cin, Cout = rng.choice([64, 128, 192, 256, 512]), rng.choice([64, 128, 256, 384, 512])
        g = torch.Generator().manual_seed(case)
        x = torch.randn(B, Cin, H, W, generator=g).to(dt)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).to(dt)
        b = torch.randn(Cout, generator=g)
        ref = F.relu(F.conv2d(x.float(), w.float(), b, padding=1))
        if pool:
            if not ops.conv_pool_fusable(H, W, Cout):
                continue
            ref = F.max_pool2d(ref, 2)
        xp = torch.zeros(B, H + 2, W + 2, Cin, dtype=dt)
        xp[:, 1:-1, 1:-1] = x.permute(0, 2, 3, 1)
        out = torch.zeros(B, ref.shape[2] + 2, ref.shape[3] + 2, Cout, device=dev, dtype=dt)
        ops.conv3x3_relu(xp.to(dev), w.permute(0, 2, 3, 1).contiguous().to(dev), b.to(dev), out, 1, pool=pool)
        what = 'conv3x3'
    got = out[:, 1:-1, 1:-1].float().cpu().permute(0, 3, 1, 2)
    tol = 4e-2 if dt == torch.bfloat16 else 6e-3
    err = float((got - ref).abs().max() / max(float(ref.abs().max()), 1e-6))
    border = float(out[:, 0].abs().max()) + float(out[:, -1].abs().max()) + float(out[:, :, 0].abs().max()) + float(out[:, :, -1].abs().max())
    ok = err <= tol and border == 0
    bad += not ok
    print('%-11s %-8s B=%d %3dx%3d %3d->%3d pool=%d rel err %.1e %s' % (what, str(dt)[6:], B, H, W, Cin, Cout, pool, err, '' if ok else '  <-- FAIL'), flush=True)
print('%d cases, %d failures' % (cases, bad))
sys.exit(1 if bad else 0)
