"""conv3_2 / conv4_2 (spatial kernel) and a long-K ping-pong GEMM for contrast; run under rocprofv3 --pmc."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgg_amd import ops
dev, dt, B = 'cuda:0', torch.bfloat16, 8
for name, H, Cin, Cout in [('conv2_2', 296, 128, 128), ('conv3_2', 148, 256, 256), ('conv4_2', 74, 512, 512)]:
    x = torch.randn(B, H + 2, H + 2, Cin, device=dev).relu().to(dt)
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) / (3 * Cin ** 0.5)).to(dt)
    b = torch.randn(Cout, device=dev)
    y = torch.empty(B, H + 2, H + 2, Cout, device=dev, dtype=dt)
    for _ in range(5):
        ops.conv3x3_relu(x, w, b, y, 1)
torch.cuda.synchronize()
A = torch.randn(7936, 4096, device=dev).relu().to(dt)
W = (torch.randn(4096, 4096, device=dev) / 64).to(dt)
out = torch.empty(7936, 4096, device=dev, dtype=dt)
for _ in range(5):
    ops.gemm(A, W, None, out=out)
torch.cuda.synchronize()
