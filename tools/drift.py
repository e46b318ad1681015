import os, sys, time, torch
sys.path.insert(0, '.')
import sgg_amd
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
from sgg_amd.rel_model_base import to_device_with_mirror
from sgg_amd.trainer import Trainer
dev='cuda:0'
bs=[]
for seed in (1,2,3):
    b = list(synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=seed))
    b[0] = [im.to(dev) for im in b[0]]
    b[3], b[4], b[5] = b[3].to(dev), to_device_with_mirror(b[4], dev), to_device_with_mirror(b[5], dev)
    bs.append(tuple(b))
torch.manual_seed(7)
model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(dev)
tr = Trainer(model, lr=1e-3, pipeline=True)
for i in range(10): tr.step(bs[i%3])
tr.flush(); torch.cuda.synchronize()
for chunk in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    t0=time.time()
    for i in range(100): tr.step(bs[i%3])
    tr.flush(); torch.cuda.synchronize()
    print('steps %4d-%4d: %.3f ms/step' % (chunk*100, chunk*100+99, (time.time()-t0)*10), flush=True)
