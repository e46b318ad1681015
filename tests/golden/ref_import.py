"""Import helper used ONLY by tests/golden/make_golden.py, in the build container.

The reference (/root/reference) needs torchvision / h5py / wandb at import time
(config.py:14-21, lib/pytorch_misc.py:6,14, sgg_models/rel_model_base.py:16-19) and none of
them is installed here.  We register *empty* stand-in modules so that the reference's own
hot-path code (message_pass, predict, UnionBoxesAndFeats, proposal_assignments_gtbox,
get_rel_inds, filter_dets, losses) can be imported and executed to produce golden vectors.
No torchvision arithmetic is emulated except `box_iou` (six lines, its documented formula),
which the reference only uses for the `> 0` overlap test (rel_model_base.py:153).

The Cython raster (lib/draw_rectangles/draw_rectangles.pyx) is built by
oracle/build_ref.py into oracle/_ref/ and injected under the module name the reference
imports (lib/get_union_boxes.py:30).
"""
import importlib.util
import os
import sys
import types

import torch

REF = os.environ.get('SGG_REFERENCE', '/root/reference')
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    return m


def _box_iou(a, b):
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = torch.max(a[:, None, :2], b[None, :, :2])
    rb = torch.min(a[:, None, 2:], b[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter / (area_a[:, None] + area_b[None] - inter)


def install():
    if not os.path.isdir(REF):
        raise RuntimeError('reference tree %s not present (golden vectors are made in the build container only)' % REF)
    if 'torchvision' not in sys.modules:
        class _Missing(object):
            def __init__(self, *a, **k):
                raise RuntimeError('torchvision is not installed; this symbol is a stand-in')
        tv = _mod('torchvision')
        tv.ops = _mod('torchvision.ops', roi_align=_Missing, MultiScaleRoIAlign=_Missing)
        tv.ops.boxes = _mod('torchvision.ops.boxes', box_iou=_box_iou)
        tv.models = _mod('torchvision.models', vgg16=_Missing)
        tv.models.detection = _mod('torchvision.models.detection', FasterRCNN=_Missing)
        tv.models.detection.faster_rcnn = _mod('torchvision.models.detection.faster_rcnn',
                                               TwoMLPHead=_Missing, FastRCNNPredictor=_Missing)
        tv.models.detection.rpn = _mod('torchvision.models.detection.rpn', AnchorGenerator=_Missing)
        tv.transforms = _mod('torchvision.transforms')
    if 'h5py' not in sys.modules:
        _mod('h5py')
    # compiled reference raster (built from the .pyx where it lies, output in oracle/_ref/)
    so_dir = os.path.join(ROOT, 'oracle', '_ref')
    cands = [f for f in os.listdir(so_dir) if f.startswith('draw_rectangles') and f.endswith('.so')] \
        if os.path.isdir(so_dir) else []
    if not cands:
        raise RuntimeError('run `python oracle/build_ref.py` first')
    spec = importlib.util.spec_from_file_location('draw_rectangles', os.path.join(so_dir, cands[0]))
    dr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dr)
    _mod('lib.draw_rectangles')
    sys.modules['lib.draw_rectangles.draw_rectangles'] = dr
    if REF not in sys.path:
        sys.path.insert(0, REF)
    argv, sys.argv = sys.argv, sys.argv[:1]
    try:
        import config  # noqa: F401  (module constants only; ModelConfig is never instantiated)
    finally:
        sys.argv = argv
    import lib
    lib.draw_rectangles = sys.modules['lib.draw_rectangles']
    return dr
