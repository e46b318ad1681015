"""IMP step kernels (sliced vs node-centric) at B = 8 and 128 images of 32 boxes: us per launch, algorithmic GB/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import sgg_amd  # noqa: E402
from sgg_amd.synthetic import SyntheticData, init_weights  # noqa: E402

model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to('cuda:0').eval()
for dtype, s in ((torch.bfloat16, 2), (torch.float32, 4)):
    model.set_compute_dtype(dtype)
    for B in [int(b) for b in os.environ.get("IMP_B", "8,32,128").split(",")]:
        algo = (2.0 * (992 * B + 32 * B) * 512) * s + 8.0 * 992 * B
        row = []
        for kind in ('sliced', 'fused'):
            ms = bench.imp_iter_ms(model, B, dtype, kind=kind)
            row.append('%s %7.2f us %6.0f GB/s' % (kind, ms * 1e3, algo / ms / 1e6))
        print(str(dtype).split('.')[-1], 'B=%3d' % B, ' | '.join(row))
