"""A short run of the randomised parity sweep (tools/fuzz_kernels.py): random ragged graphs through both IMP step kernels, random
GEMM / TN-GEMM shapes, 3x3 convs (+ fused pool), RoIAlign incl. degenerate boxes, both rasters, pair indexing -- against dense torch
fp32 / the oracle.  The long form is `python tools/fuzz_kernels.py 120` (~280 cases of each per 100 s, all passing in round 1)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_fuzz_sweep_short():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import fuzz_kernels
    out = fuzz_kernels.run(budget=10.0, seed=12345)
    assert min(out.values()) >= 3, out
