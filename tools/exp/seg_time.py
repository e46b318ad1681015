"""Wall time of nested prefixes of the train step (steady state, no host syncs inside the loops)."""
import sys, time, torch
sys.path.insert(0, '.')
import sgg_amd
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
def timeit(fn, n=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
tr = Trainer(model, lr=1e-3, pipeline=False)
model.train()
def vgg_only():
    with torch.no_grad():
        r = model.faster_rcnn(b[0], b[3], b[4], b[5])
def feats():
    with torch.no_grad():
        r = model.faster_rcnn(b[0], b[3], b[4], b[5])
        ri = r.rel_labels[:, :3].contiguous()
        rois = torch.cat((r.im_inds[:, None].float(), r.rm_box_priors), 1)
        model.node_edge_features(r.fmap, rois, ri[:, 1:], r.im_sizes)
def fwd():
    with torch.no_grad():
        model([b])
def fwd_grad():
    res = model([b])
    return res
def fwd_bwd():
    res = model([b])
    loss = tr.losses(res)
    tr.opt.zero_grad()
    loss.backward()
print('detector (prep + VGG + labels)   %.3f ms' % timeit(vgg_only))
print('  + RoIAlign                      %.3f ms' % timeit(feats))
print('  + head forward (no grad graph)  %.3f ms' % timeit(fwd))
print('  + autograd graph                %.3f ms' % timeit(fwd_grad))
print('  + losses + backward             %.3f ms' % timeit(fwd_bwd))
print('  + clip + SGD (in order)         %.3f ms' % timeit(lambda: tr.step(b)))
tr2 = Trainer(model, lr=1e-3, pipeline=True)
print('  pipelined optimiser             %.3f ms' % timeit(lambda: tr2.step(b)))
# ---- does the optimiser really run UNDER the detector?
from sgg_amd.imp import node_lane
side = node_lane(torch.device(dev))[0]
res = model([b]); loss = tr.losses(res); tr.opt.zero_grad(); loss.backward(); torch.cuda.synchronize()
grads = {p: p.grad for p in tr.opt.params() if p.grad is not None}
tr.opt.max_blocks = 256
def sgd_side():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        tr.opt.step(grads=grads)
    torch.cuda.current_stream().wait_stream(side)
def both():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        tr.opt.step(grads=grads)
    vgg_only()
    torch.cuda.current_stream().wait_stream(side)
print('optimiser alone (side stream)     %.3f ms' % timeit(sgd_side))
print('detector alone                    %.3f ms' % timeit(vgg_only))
print('optimiser || detector             %.3f ms' % timeit(both))
for mb in (64, 128, 512, 0):
    tr.opt.max_blocks = mb
    print('  max_blocks=%4d: alone %.3f  with detector %.3f' % (mb, timeit(sgd_side), timeit(both)))
