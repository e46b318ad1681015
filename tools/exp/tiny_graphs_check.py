import sys, torch, numpy as np
sys.path.insert(0, '.')
import sgg_amd
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
from sgg_amd.trainer import Trainer
from oracle import sgg_oracle as O
dev = 'cuda:0'
S = 96
model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls', min_size=S, max_size=S)).to(dev)
for nb in (1, 2, 3):
    for dtype in (torch.float32, torch.bfloat16):
        model.set_compute_dtype(dtype)
        model.eval()
        b = synthetic_batch(B=2, S=S, n_boxes=nb, n_fg=min(1, nb * (nb - 1)), seed=nb)
        out = model([tuple(b)])
        print('eval', nb, dtype, [o.shape for o in out], np.isfinite(out[4]).all())
        if nb >= 2:
            tr = Trainer(model, lr=1e-3)
            l = [float(tr.step(tuple(b))) for _ in range(2)]
            tr.flush()
            print('train', nb, dtype, l)
print('ok')
