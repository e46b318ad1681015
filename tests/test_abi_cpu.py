"""CPU: the C-ABI library builds/loads here (hipcc cross-compiles gfx950 without a GPU) and exports every symbol that
include/sgg_hip.h declares; the Python mirror has the reference's surface.  No compute calls (no GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, 'include', 'sgg_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(sgg_[a-z0-9_]+)\s*\(', txt)))


def test_header_symbols_all_exported_and_typed():
    from sgg_amd import _lib
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), s
        assert s in _lib.SIGNATURES, 'ctypes signature missing for %s' % s
    assert sorted(_lib.SIGNATURES) == syms
    assert lib.sgg_abi_version() == _lib.ABI_VERSION
    assert b'gfx950' in lib.sgg_build_info()


def test_argument_counts_match_header():
    from sgg_amd import _lib
    txt = open(os.path.join(ROOT, 'include', 'sgg_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    for name, args in re.findall(r'\b(sgg_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;', txt, flags=re.S):
        n = 0 if args.strip() in ('', 'void') else len(args.split(','))
        assert n == len(_lib.SIGNATURES[name]), (name, n, len(_lib.SIGNATURES[name]))


def test_library_is_gfx950_code_object():
    from sgg_amd import _lib
    blob = open(_lib.LIB_PATH, 'rb').read()
    # every embedded device code object (offload-bundle entry id "...amdhsa--gfxNNN") targets gfx950 only
    # (rocPRIM's host-side arch-name table also mentions other gfx names; those are strings, not code)
    targets = set(re.findall(rb'amdhsa--(gfx[0-9a-z]+)', blob))
    assert targets == {b'gfx950'}, targets


def test_product_path_never_imports_oracle():
    pkg = os.path.join(ROOT, 'sgg_amd')
    for f in os.listdir(pkg):
        if f.endswith('.py'):
            src = open(os.path.join(pkg, f)).read()
            assert 'oracle' not in re.sub(r'""".*?"""', '', src, flags=re.S).replace('# oracle', ''), f


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from sgg_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(ImportError):
        _lib.load()


def test_cpu_tensor_is_rejected_not_silently_computed():
    import torch
    from sgg_amd import ops
    with pytest.raises(RuntimeError):
        ops.gemm(torch.zeros(64, 64), torch.zeros(64, 64))


def test_model_surface_matches_reference():
    import torch
    import sgg_amd
    from sgg_amd.synthetic import SyntheticData
    m = sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')
    for attr in ('detector', 'edge_dim', 'pool_sz', 'fmap_sz', 'mode', 'roi_fmap', 'roi_fmap_obj', 'union_boxes',
                 'obj_dim', 'hidden_dim', 'mp_iter', 'RELS_PER_IMG', 'require_overlap', 'num_classes', 'num_rels'):
        assert hasattr(m, attr), attr
    for meth in ('forward', 'predict', 'message_pass', 'node_edge_features', 'get_scaled_boxes', 'set_box_score_thresh',
                 'get_rel_inds', 'faster_rcnn', 'gt_labels'):
        assert callable(getattr(m, meth)), meth
    assert (m.edge_dim, m.pool_sz, m.fmap_sz, m.obj_dim) == (512, 7, 38, 4096)
    trainable = sum(p.numel() for n, p in m.named_parameters() if not n.startswith('detector.'))
    assert trainable == 247753678          # DP gradient payload
    # LR groups of lib/pytorch_misc.py:135-136 key on the 'roi_fmap' prefix
    assert any(n.startswith('roi_fmap') for n, _ in m.named_parameters())
    with pytest.raises(NotImplementedError):
        sgg_amd.RelModelStanford(SyntheticData(), backbone='vgg16_old')
    r = sgg_amd.Result(rm_obj_dists=torch.zeros(1), rel_labels=None)
    assert hasattr(r, 'rm_obj_dists') and not hasattr(r, 'rel_labels')     # lib/pytorch_misc.py:696-700


def test_blob_tuple_layout_matches_reference_contract():
    """dataloaders/blob.py:77-126,145-166,214-261 for the one-device case (the 8/9-tuple the model's forward unpacks)."""
    import numpy as np
    import torch
    from sgg_amd.blob import Blob, vg_collate
    rng = np.random.RandomState(0)
    data = []
    for i, nb in enumerate((3, 5)):
        data.append({'img': rng.randint(0, 255, size=(40 + i, 50, 3)).astype(np.uint8), 'img_size': (592, 592, 1.0),
                     'gt_boxes': rng.rand(nb, 4) * 100, 'gt_classes': rng.randint(1, 151, nb), 'scale': 0.5,
                     'gt_relations': np.array([[0, 1, 7], [2, 0, 3]]), 'fn': '/some/dir/%d.jpg' % i})
    blob = vg_collate(data, num_gpus=1, is_train=True, mode='rel')
    t = blob[0]
    assert len(t) == 9 and len(blob) == 1 and t[2] == 0 and t[7] is None and t[8] == ['0.jpg', '1.jpg']
    assert t[1].shape == (2, 3) and t[3].dtype == torch.float32 and tuple(t[3].shape) == (8, 4)
    np.testing.assert_allclose(t[3][:3].numpy(), (data[0]['gt_boxes'].astype(np.float32) * 0.5))
    assert t[4].dtype == torch.int64 and t[4][:, 0].tolist() == [0, 0, 0, 1, 1, 1, 1, 1]
    assert t[5].tolist() == [[0, 0, 1, 7], [0, 2, 0, 3], [1, 0, 1, 7], [1, 2, 0, 3]] and t[6] is None
    ev = vg_collate(data, is_train=False, mode='rel')[0]
    assert len(ev) == 8 and ev[7] == ['0.jpg', '1.jpg']
    with pytest.raises(ValueError):
        b = Blob(mode='rel', batch_size_per_gpu=3)
        b.append(data[0])
        b.reduce()
    with pytest.raises(ValueError):
        Blob(num_gpus=2)
    with pytest.raises(ValueError):
        blob[1]


def test_abi_version_is_bumped_with_the_prototypes():
    """ADVICE r3 / VERDICT r3 item 12: a prototype change must come with a new SGG_ABI_VERSION, otherwise a stale
    libsgg_hip.so passes load()'s check and is called with shifted arguments.  tests/abi.lock (written by
    tools/abi_lock.py, which never re-points a recorded version) pins version -> digest of all prototypes."""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import abi_lock
    from sgg_amd import _lib
    lock = json.load(open(os.path.join(ROOT, 'tests', 'abi.lock')))
    v = abi_lock.header_version()
    assert v == _lib.ABI_VERSION == _lib.load().sgg_abi_version()
    assert lock.get(str(v)) == abi_lock.digest(), \
        'prototypes of include/sgg_hip.h changed: bump SGG_ABI_VERSION (header + _lib.py), then run tools/abi_lock.py'
    assert v == max(int(k) for k in lock)


def test_stale_library_version_is_refused(monkeypatch):
    from sgg_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'ABI_VERSION', _lib.ABI_VERSION + 1)
    with pytest.raises(ImportError, match='ABI version'):
        _lib.load()


def test_integration_md_stubs_match_the_bound_signatures():
    """Every python code block of INTEGRATION.md that binds or calls a C symbol does so with the header's argument
    list: `lib.<sym>.argtypes = [...]` equals _lib.SIGNATURES[sym], `lib.<sym>(...)` passes that many arguments."""
    import ast
    import ctypes
    from sgg_amd import _lib
    md = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', md, flags=re.S)
    assert blocks
    bound, called = set(), set()
    for src in blocks:
        try:
            tree = ast.parse(src)
        except SyntaxError:                    # prose-like excerpts ("..." lines) bind nothing
            tree = ast.parse(src.replace('\n...\n', '\n'))
        for node in ast.walk(tree):
            if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Attribute) \
                    and node.targets[0].attr == 'argtypes' and isinstance(node.targets[0].value, ast.Attribute):
                sym = node.targets[0].value.attr
                got = eval(compile(ast.Expression(node.value), 'INTEGRATION.md', 'eval'), {'ctypes': ctypes})
                assert got == _lib.SIGNATURES[sym], (sym, got, _lib.SIGNATURES[sym])
                bound.add(sym)
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr.startswith('sgg_') \
                    and node.func.attr in _lib.SIGNATURES:
                assert len(node.args) == len(_lib.SIGNATURES[node.func.attr]), \
                    (node.func.attr, len(node.args), len(_lib.SIGNATURES[node.func.attr]))
                called.add(node.func.attr)
            if isinstance(node, ast.Compare) and isinstance(node.left, ast.Call) \
                    and getattr(node.left.func, 'attr', '') == 'sgg_abi_version':
                assert node.comparators[0].value == _lib.ABI_VERSION
    assert 'sgg_union_rects_fwd' in bound and 'sgg_union_rects_fwd' in called


def test_import_order_torch_is_loaded_before_the_library():
    """`import sgg_amd` as the FIRST import of a process (what `from sgg_amd import RelModelStanford` at the top of main.py is): torch -- and
    with it the HIP runtime its wheel bundles -- must be in the process before libsgg_hip.so is loaded, or the library binds the system
    ROCm's libamdhip64 and the two runtimes do not share streams / modules (round 6: the first launch failed)."""
    import subprocess
    import sys
    code = ("import sys\n"
            "import sgg_amd\n"
            "import ctypes\n"
            "assert 'torch' in sys.modules\n"
            "maps = open('/proc/self/maps').read()\n"
            "hips = sorted({l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l})\n"
            "print(hips)\n"
            "assert len(hips) <= 1, hips\n")
    out = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
