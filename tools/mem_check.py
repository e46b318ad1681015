import os, sys, time, torch
sys.path.insert(0, '.')
import sgg_amd
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
from sgg_amd.rel_model_base import to_device_with_mirror
from sgg_amd.trainer import Trainer
dev='cuda:0'
model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(dev)
bs=[]
for seed in (1,2,3):
    b = list(synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=seed))
    b[0] = [im.to(dev) for im in b[0]]
    b[3], b[4], b[5] = b[3].to(dev), to_device_with_mirror(b[4], dev), to_device_with_mirror(b[5], dev)
    bs.append(tuple(b))
tr = Trainer(model, lr=1e-3, pipeline=True)
for mode in ('three batches', 'one batch'):
    for i in range(20):
        tr.step(bs[i % 3 if mode[0] == 't' else 0])
    tr.flush(); torch.cuda.synchronize()
    t0=time.time()
    for i in range(200):
        loss = tr.step(bs[i % 3 if mode[0] == 't' else 0])
    tr.flush(); torch.cuda.synchronize()
    print(mode, 'loss %.4f' % float(loss), 'alloc %.2f GB reserved %.2f GB' % (torch.cuda.memory_allocated()/2**30, torch.cuda.memory_reserved()/2**30), '%.3f ms/step' % ((time.time()-t0)/200*1e3), flush=True)
for b in bs:
    res = model([b]); print('plain', getattr(res.rel_labels, '_sgg_plain', None), 'rows', res.rel_labels.shape[0])
