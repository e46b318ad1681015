"""A short run of the randomised parity sweep (tools/fuzz_kernels.py): random ragged graphs through both IMP step kernels, random
GEMM / TN-GEMM shapes, against dense torch fp32.  The long form is `python tools/fuzz_kernels.py 120` (614 + 617 + 617 cases pass)."""
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
    out = fuzz_kernels.run(budget=8.0, seed=12345)
    assert min(out.values()) >= 5, out
