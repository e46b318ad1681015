"""IMP step kernels at B = 8 / 32 / 128 images of 32 boxes: us per launch and algorithmic GB/s for the short-lived
one-workgroup-per-unit kernel (what the forward runs), the split step (write stream + LDS-DMA read stream), the register-prefetch
persistent kernel and the node-centric kernel.  IMP_B=8,128 selects batch sizes; IMP_FORMS=0,s,1,fused selects forms."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import sgg_amd  # noqa: E402
from sgg_amd.synthetic import SyntheticData, init_weights  # noqa: E402

model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to('cuda:0').eval()
names = {'s': 'split', '1': 'stream', '0': 'short', 'fused': 'node-centric'}
forms = os.environ.get('IMP_FORMS', '0,s,1,fused').split(',')
for dtype, s in ((torch.bfloat16, 2), (torch.float32, 4)):
    model.set_compute_dtype(dtype)
    for B in [int(b) for b in os.environ.get('IMP_B', '8,32,128').split(',')]:
        algo = (2.0 * (992 * B + 32 * B) * 512) * s + 8.0 * 992 * B
        row = []
        for form in forms:
            if form != 'fused':
                os.environ['SGG_IMP_STREAM'] = form
            ms = bench.imp_iter_ms(model, B, dtype, kind='fused' if form == 'fused' else 'sliced')
            row.append('%s %7.2f us %5.0f GB/s %.3f' % (names[form], ms * 1e3, algo / ms / 1e6, algo / ms / 1e6 / 8000))
        os.environ.pop('SGG_IMP_STREAM', None)
        print(str(dtype).split('.')[-1], 'B=%3d' % B, ' | '.join(row), flush=True)
