"""The IMP step's launch (sgg_imp_ctx_fwd) at B = 8 ... 512 images of 32 boxes: us per launch and GB/s of SURVEY 8(d)'s algorithmic bytes for
each form (sliced / lists, routed default), and the gate kernel that takes the place of the edge inputs.  IMP_B=8,128 selects batch sizes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import sgg_amd  # noqa: E402
from sgg_amd.synthetic import SyntheticData, init_weights  # noqa: E402

model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to('cuda:0').eval()
kinds = os.environ.get('IMP_FORMS', 'ctx,ctx_sliced,ctx_mfma,ctx_lists,gate_proj').split(',')
for dtype, s in ((torch.bfloat16, 2), (torch.float16, 2), (torch.float32, 4)):
    model.set_compute_dtype(dtype)
    for B in [int(b) for b in os.environ.get('IMP_B', '8,32,128,512').split(',')]:
        algo = (2.0 * (992 * B + 32 * B) * 512) * s + 8.0 * 992 * B
        row = []
        for kind in kinds:
            if kind == 'ctx_mfma' and dtype == torch.float32:
                continue
            ms = bench.imp_iter_ms(model, B, dtype, kind=kind)
            row.append('%s %7.2f us %5.0f GB/s %.3f' % (kind, ms * 1e3, algo / ms / 1e6, algo / ms / 1e6 / 8000))
        print(str(dtype).split('.')[-1], 'B=%3d' % B, ' | '.join(row), flush=True)
