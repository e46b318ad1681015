"""Where does the host->HBM hop spend its time?  python tools/stager_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgg_amd  # noqa: E402
from sgg_amd import ops  # noqa: E402
from sgg_amd.blob import DeviceStager  # noqa: E402


def t(fn, n=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


B, S = 8, 592
u8 = [torch.randint(0, 256, (S, S, 3), dtype=torch.uint8) for _ in range(B)]
f32 = [torch.rand(3, S, S) for _ in range(B)]
batch = (u8, None, 0, torch.rand(256, 4), torch.zeros(256, 2, dtype=torch.int64), torch.zeros(48, 4, dtype=torch.int64), None, [])
st = DeviceStager()
print('stage (pack + async copy + wait)   %.3f ms' % t(lambda: st.stage(batch)))
pinned = torch.empty(B * S * S * 3, dtype=torch.uint8).pin_memory()
print('host pack 8 x 1 MB into pinned     %.3f ms' % t(lambda: [pinned[i * S * S * 3:(i + 1) * S * S * 3].copy_(u8[i].reshape(-1)) for i in range(B)]))
dev = torch.empty_like(pinned, device='cuda')
print('one 8.4 MB pinned H2D              %.3f ms' % t(lambda: dev.copy_(pinned, non_blocking=True)))
print('8 pageable f32 .to(cuda)           %.3f ms' % t(lambda: [im.to('cuda', non_blocking=True) for im in f32]))
buf = torch.zeros(B, 610, 610, 4, device='cuda')
du8 = [im.cuda() for im in u8]
df32 = [im.cuda() for im in f32]
print('image_prep_u8 x8                   %.3f ms' % t(lambda: [ops.image_prep_u8(du8[b], S, S, buf, b) for b in range(B)]))
print('image_prep f32 x8                  %.3f ms' % t(lambda: [ops.image_prep(df32[b], S, S, buf, b) for b in range(B)]))

# ---- whole forward from host batches
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch  # noqa: E402
model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to('cuda').eval()
model.set_compute_dtype(torch.bfloat16)
sb = list(synthetic_batch(B=B, S=S, n_boxes=32, n_fg=6, seed=1))
host = tuple(sb)
u8b = list(sb)
u8b[0] = [(im * 255).round().to(torch.uint8).permute(1, 2, 0).contiguous() for im in sb[0]]
u8b = tuple(u8b)
res = list(sb)
res[0] = [im.cuda() for im in sb[0]]
res[3], res[4], res[5] = sb[3].cuda(), sb[4].cuda(), sb[5].cuda()
res = tuple(res)
with torch.no_grad():
    print('forward, resident inputs           %.3f ms' % t(lambda: model([res]), 10))
    print('forward, host f32                  %.3f ms' % t(lambda: model([host]), 10))
    print('forward, stage(u8)                 %.3f ms' % t(lambda: model([st.stage(u8b)]), 10))
    staged = st.stage(u8b)
    print('forward, pre-staged u8             %.3f ms' % t(lambda: model([staged]), 10))

    def loop():
        for b_ in st.prefetch([u8b] * 10):
            model([b_])
    print('prefetch loop / 10                 %.3f ms' % (t(loop, 2) / 10))

