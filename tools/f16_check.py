"""f16 / bf16 mode at the bench configuration: logits against the f32 mode of the same kernels (itself within 8e-5 of the oracle), a few
train steps (loss falls, finite), and the two modes' step times."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgg_amd
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
from sgg_amd.rel_model_base import to_device_with_mirror
from sgg_amd.trainer import Trainer
DEV = 'cuda:0'
model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(DEV).eval()
batch = list(synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=111))
batch[0] = [im.to(DEV) for im in batch[0]]
batch[3], batch[4], batch[5] = batch[3].to(DEV), to_device_with_mirror(batch[4], DEV), to_device_with_mirror(batch[5], DEV)
batch = tuple(batch)

def logits(dt):
    model.set_compute_dtype(dt)
    with torch.no_grad():
        res = model.faster_rcnn(batch[0], batch[3], batch[4], None)
        rel_inds = model.get_rel_inds(None, res.im_inds, res.rm_box_priors)
        rois = torch.cat((res.im_inds[:, None].float(), res.rm_box_priors), 1)
        nf, ef = model.node_edge_features(res.fmap, rois, rel_inds[:, 1:], res.im_sizes)
        od, rd = model.predict(nf, ef, rel_inds, rois, res.im_sizes, _im_inds=res.im_inds.contiguous())
    return od.float(), rd.float()
ref = logits(torch.float32)
for name, dt in (('f16', torch.float16), ('bf16', torch.bfloat16)):
    od, rd = logits(dt)
    print('%s vs f32 mode: obj max %.4f mean %.5f | rel max %.4f mean %.5f | finite %s' % (
        name, float((od - ref[0]).abs().max()), float((od - ref[0]).abs().mean()), float((rd - ref[1]).abs().max()), float((rd - ref[1]).abs().mean()),
        bool(torch.isfinite(od).all() and torch.isfinite(rd).all())), flush=True)
for name, dt in (('f16', torch.float16), ('bf16', torch.bfloat16)):
    m = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(DEV)
    m.set_compute_dtype(dt)
    tr = Trainer(m, lr=1e-3, pipeline=True)
    losses = [float(tr.step(batch)) for _ in range(12)]
    tr.flush()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        tr.step(batch)
    tr.flush()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    finite = all(bool(torch.isfinite(p).all()) for p in m.parameters())
    print('%s train: loss %.4f -> %.4f, grad norm %.4f, params finite %s, %.3f ms/step = %.1f images/s' % (
        name, losses[0], losses[-1], tr.opt.grad_norm(1.0 / tr.loss_scale), finite, ms, 8e3 / ms), flush=True)
    m.eval()
    torch.cuda.synchronize()
    with torch.no_grad():
        for _ in range(3):
            m([batch])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            m([batch])
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print('%s infer: %.3f ms/step = %.1f images/s' % (name, ms, 8e3 / ms), flush=True)
    del tr, m
