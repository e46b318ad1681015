"""Per-image eval latency (the reference evaluates one image per call, dataloaders/visual_genome.py:730)."""
import sys, time, torch, cProfile, pstats
sys.path.insert(0, '.')
import sgg_amd
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
from sgg_amd.rel_model_base import to_device_with_mirror
dev = 'cuda:0'
model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(dev).eval()
model.set_compute_dtype(torch.bfloat16)
for nb in (8, 20, 32):
    b = list(synthetic_batch(B=1, S=592, n_boxes=nb, n_fg=4, seed=3))
    b[0] = [im.to(dev) for im in b[0]]
    b[3], b[4], b[5] = b[3].to(dev), to_device_with_mirror(b[4], dev), to_device_with_mirror(b[5], dev)
    b = tuple(b)
    with torch.no_grad():
        for _ in range(5):
            model([b])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            model([b])
        dt = (time.perf_counter() - t0) / 30
    print('boxes %2d: %.3f ms per image (%.0f images/s)' % (nb, 1e3 * dt, 1 / dt))
pr = cProfile.Profile(); pr.enable()
with torch.no_grad():
    for _ in range(10):
        model([b])
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(12)
