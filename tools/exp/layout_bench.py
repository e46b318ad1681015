"""boxes_to_layout at the GAN's size (8 images x 32 objects, 512 channels, 7x7 -> 38x38): kernel time and HBM roofline fraction.
Algorithmic bytes = patches read once + layout written once (fwd); layout gradient read once + patch gradients written once (bwd).
    python tools/exp/layout_bench.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sgg_amd.gan_ops import boxes_to_layout_nhwc
dev = 'cuda:0'
rng = np.random.RandomState(5)
B, nb, D, S, H = 8, 32, 512, 7, 38
xy = rng.uniform(0, 0.7, size=(B * nb, 2)); wh = rng.uniform(0.03, 0.4, size=(B * nb, 2))
boxes = torch.from_numpy(np.concatenate((xy, np.minimum(xy + wh, 1.0)), 1).astype(np.float32)).to(dev)
img = torch.from_numpy(np.repeat(np.arange(B), nb)).to(dev)
for dtype in (torch.float32, torch.bfloat16):
    v = torch.randn(B * nb, S, S, D, device=dev).to(dtype).requires_grad_(True)
    u = torch.randn(B, H, H, D, device=dev).to(dtype)
    es = v.element_size()
    alg = (v.numel() + B * H * H * D) * es
    def timeit(fn, n=50):
        for _ in range(5): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n
    with torch.no_grad():
        t_f = timeit(lambda: boxes_to_layout_nhwc(v, boxes, img, H, H, 'sum', num_images=B))
    out = boxes_to_layout_nhwc(v, boxes, img, H, H, 'sum', num_images=B)
    def bw():
        v.grad = None
        out.backward(u, retain_graph=True)
    t_b = timeit(bw)
    print('%s  fwd %.1f us (%.0f GB/s, %.2f of 8 TB/s)   bwd %.1f us (%.0f GB/s)   algorithmic %.1f MB;  the reference materialises %.0f MB of per-object canvases'
          % (str(dtype).split('.')[-1], t_f * 1e3, alg / t_f / 1e6, alg / t_f / 1e6 / 8000, t_b * 1e3, alg / t_b / 1e6, alg / 1e6, B * nb * D * H * H * es / 1e6))
