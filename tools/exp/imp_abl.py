import os, sys, torch
sys.path.insert(0, '.')
import bench, sgg_amd
from sgg_amd.synthetic import SyntheticData, init_weights
model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to('cuda:0').eval()
model.set_compute_dtype(torch.bfloat16)
print('LP', os.environ.get('SGG_IMP_LP', 'auto'), ' '.join('B=%d %.2f us' % (B, 1e3 * bench.imp_iter_ms(model, B, torch.bfloat16, kind='sliced')) for B in (8, 32, 128)))
