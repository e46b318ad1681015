"""How long does the head forward wait for the queued optimiser update + operand rebuild?  (timing only: variants b/c train wrongly)"""
import sys, time, torch
sys.path.insert(0, '.')
import sgg_amd
from sgg_amd import trainer as TR
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
from sgg_amd.trainer import Trainer
from sgg_amd.rel_model_base import to_device_with_mirror
dev = 'cuda:0'
model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(dev)
model.set_compute_dtype(torch.bfloat16)
b = list(synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=111))
b[0] = [im.to(dev) for im in b[0]]
b[3], b[4], b[5] = b[3].to(dev), to_device_with_mirror(b[4], dev), to_device_with_mirror(b[5], dev)
b = tuple(b)
def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
tr = Trainer(model, lr=1e-4, pipeline=True)
print('a) pipelined, as shipped                    %.3f ms' % timeit(lambda: tr.step(b)))
import sgg_amd.train as T
real_tw = T.train_weights
cached = real_tw(model)
T.train_weights = lambda m: cached
tr._bump = lambda: None
tr.opt.on_update = None
print('b) no operand rebuild after the update      %.3f ms' % timeit(lambda: tr.step(b)))
real_step = tr.opt.step
tr.opt.step = lambda *a, **k: None
print('c) no optimiser kernels either              %.3f ms' % timeit(lambda: tr.step(b)))
