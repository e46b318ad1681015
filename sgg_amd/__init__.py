"""sgg_amd -- MI355X-native hot path of bknyaz/sgg (Faster-R-CNN features + Stanford IMP relation model).

Host side in Python (torch for device memory / streams / torch.distributed), numerics in libsgg_hip.so
(hand-written HIP for gfx950, C ABI in include/sgg_hip.h).  Importing this package loads the library and
raises ImportError if it has not been built: there is no CPU or torch fallback.
"""
import os as _os

# HIP streams are multiplexed onto GPU_MAX_HW_QUEUES hardware queues (default 4).  With RCCL's streams in the process the
# side stream of the node lane / pipelined optimiser landed on the SAME queue as the main stream and every overlap was
# silently serialised (measured: 11.6 vs 10.2 ms per data-parallel step).  Must be set before the HIP runtime starts.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

# torch FIRST: its wheel bundles its own HIP runtime (libamdhip64), and libsgg_hip.so must bind to THAT copy -- whichever libamdhip64 is
# loaded first serves both.  With `import sgg_amd` as a process's first import the library used to pull in the system ROCm's runtime
# ahead of torch's: two runtimes in one process, and the first kernel launch failed (found in round 6 by running build() and smoke() in
# one process; tests/test_abi_cpu.py::test_import_order).
import torch as _torch  # noqa: E402,F401

from . import _lib  # noqa: E402

_lib.load()

from .result import Result  # noqa: E402,F401
from .rel_model_base import RelModelBase  # noqa: E402,F401
from .rel_model_stanford import RelModelStanford  # noqa: E402,F401

__all__ = ['Result', 'RelModelBase', 'RelModelStanford']
