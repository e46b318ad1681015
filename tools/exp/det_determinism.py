"""Run-to-run determinism of the detector's feature map and detections (bf16 / f32), and which run agrees with fp32."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import sgg_amd
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgdet')).to('cuda:0').eval()
batch = synthetic_batch(B=4, S=592, n_boxes=8, n_fg=2, seed=41)
images = [batch[0][i] for i in range(4)]
fm = {}
for dt in (torch.float32, torch.bfloat16, torch.float32, torch.bfloat16):
    model.set_compute_dtype(dt)
    for rep in range(3):
        with torch.no_grad():
            fmap, sizes, padded = model.detector.features(images, dt)
            torch.cuda.synchronize()
            fm.setdefault(dt, []).append(fmap.float().cpu().clone())
ref = fm[torch.float32][0]
print('f32 runs identical:', all(torch.equal(ref, t) for t in fm[torch.float32]))
for i, t in enumerate(fm[torch.bfloat16]):
    print('bf16 run %d: max |x - f32| %.4f  mean %.5f   identical to bf16 run 0: %s  to run 1: %s' %
          (i, float((t - ref).abs().max()), float((t - ref).abs().mean()), torch.equal(t, fm[torch.bfloat16][0]), torch.equal(t, fm[torch.bfloat16][1])))
d = (fm[torch.bfloat16][0] - fm[torch.bfloat16][1]).abs()
print('run0 vs run1: differing elements %d of %d; where (b,y,x,c) first few:' % (int((d > 0).sum()), d.numel()), (d > 0).nonzero()[:8].tolist())
