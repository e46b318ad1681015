"""ctypes binding of libsgg_hip.so (the C ABI declared in include/sgg_hip.h).

There is NO fallback: if the shared library is missing or does not export a symbol, importing the
product path raises.  The oracle (oracle/) is never imported from here.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SGG_HIP_LIB') or os.path.join(_HERE, 'libsgg_hip.so')   # override: kernel experiments only

SGG_F32, SGG_BF16, SGG_F16 = 0, 1, 2
ACT_NONE, ACT_RELU = 0, 1
ABI_VERSION = 14

import threading  # noqa: E402
# held by sgg_amd.graph_step while it captures a hipGraph, and by every other thread of this package around its own GPU calls (the
# DeviceStager's worker: copies, event records): no call of another thread lands in the middle of a stream capture
CAPTURE_LOCK = threading.RLock()

_P, _I, _F, _L = c_void_p, c_int, c_float, c_int64

# name -> argtypes (all return int unless listed in _RESTYPE)
SIGNATURES = {
    'sgg_abi_version': [],
    'sgg_build_info': [],
    'sgg_image_prep': [_P, _I, _I, _I, _I, _P, _I, _I, _I, _P],
    'sgg_image_prep_u8': [_P, _I, _I, _I, _I, _P, _I, _I, _I, _P],
    'sgg_image_prep_batch': [_P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _P],
    'sgg_conv1_1': [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    'sgg_conv1_block': [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    'sgg_conv1_pack_weights': [_P, _P, _I, _P],
    'sgg_conv3x3_relu': [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    'sgg_conv3x3_relu_x3': [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    'sgg_transpose_multi': [_P, _P, _P, _P, _P, _P, _I, _I, _P],
    'sgg_split3': [_P, _L, _L, _I, _I, _P, _L, _I, _P],
    'sgg_split2': [_P, _L, _L, _I, _I, _P, _L, _P],
    'sgg_maxpool2x2': [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    'sgg_pair_index_eval': [_P, _P, _I, _I, _P, _I, _P, _P, _P],
    'sgg_pair_index_train': [_P, _I, _P, _I, _P, _P, _I, _P, _P, _P],
    'sgg_rel_assign_tables': [_P, _P, _P, _I, _P, _P, _I, _F, _I, _P, _P, _P, _P],
    'sgg_boxes_to_layout_fwd': [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P],
    'sgg_boxes_to_layout_bwd': [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P],
    'sgg_triple_gather': [_P, _P, _P, _I, _I, _I, _P, _I, _P],
    'sgg_triple_pool_fwd': [_P, _I, _I, _P, _P, _P, _P, _I, _I, _I, _P, _I, _P],
    'sgg_triple_pool_bwd': [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _I, _P],
    'sgg_edge_csr': [_P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P],  # rel, E, N, im_inds, out_ptr, out_ids, in_ptr, in_ids, so, flags, stream
    'sgg_roi_align_fwd': [_P, _I, _I, _I, _I, _P, _I, _P, _I, _F, _I, _I, _P, _P, _I, _P],
    'sgg_roi_align_bwd': [_P, _I, _I, _I, _I, _P, _I, _P, _I, _F, _I, _I, _P, _I, _I, _P],
    'sgg_union_rects_fwd': [_P, _P, _I, _I, _F, _P, _I, _P, _P],
    'sgg_union_rect_patches': [_P, _P, _I, _I, _P, _I, _I, _P, _I, _P],
    'sgg_max4_rows': [_P, _P, _I, _I, _I, _P],
    'sgg_bcast_add': [_P, _P, _I, _I, _I, _I, _P],
    'sgg_gemm': [_P, _I, _P, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    'sgg_gemm_splitk': [_P, _I, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    'sgg_gru_gate_fwd': [_P, _P, _P, _P, _P, _I, _I, _P, _I, _P, _I, _I, _P],
    'sgg_imp_sliced_capacity': [_I, _I],
    'sgg_imp_ctx_mfma_min_units': [],
    # x, so, out_ptr, out_ids, in_ptr, in_ids, img_ptr, B, N, E, H, node_dots, edge_dots, gate_b, pair, out, max_edges, max_nodes, sum_ctx, dtype, stream
    'sgg_imp_ctx_fwd': [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _P],
    'sgg_gru_gate_proj_fwd': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _I, _P, _I, _I, _P],
    'sgg_gru_gate_proj_bwd': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    'sgg_imp_edge_ctx_bwd': [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    'sgg_imp_node_gates_bwd': [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _P],
    'sgg_graph_ptr': [_P, _I, _I, _P, _P, _P],
    'sgg_im2col': [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _I, _I, _P],
    'sgg_upsample_add': [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    'sgg_col2im': [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P],
    'sgg_maxpool3x3s2': [_P, _P, _I, _I, _I, _I, _I, _P],
    'sgg_plane_copy': [_P, _I, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    'sgg_add_relu': [_P, _P, _L, _I, _P],
    'sgg_eval_tail': [_P, _I, _I, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    'sgg_rpn_decode': [_P, _I, _P, _I, _I, _I, _F, _F, _I, _P, _P, _P],
    'sgg_segmented_sort_desc': [_P, _P, _P, _P, _I, _I, _P, _I, _P, _P, _P],
    'sgg_gather_topk': [_P, _P, _P, _P, _P, _P, _I, _I, _F, _P, _P, _P, _P, _P],
    'sgg_topk_gather': [_P, _P, _P, _P, _P, _I, _I, _F, _P, _P, _P, _P, _P],
    'sgg_nms': [_P, _P, _P, _I, _I, _F, _I, _P, _P, _P, _P],
    'sgg_compact_rois': [_P, _P, _P, _I, _I, _I, _P, _P, _P],
    'sgg_det_candidates': [_P, _I, _P, _I, _I, _P, _F, _F, _P, _P, _P, _P],
    'sgg_det_output': [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P],
    'sgg_dropout_fwd': [_P, _L, _F, ctypes.c_uint64, _I, _P],
    'sgg_dropout_fwd_dev': [_P, _L, _F, _P, ctypes.c_uint64, _I, _P],
    'sgg_allreduce_unique_id': [_P],
    'sgg_allreduce_init': [_P, _I, _I, _P],
    'sgg_allreduce_sum': [_P, _P, _L, _I, _P],
    'sgg_allreduce_destroy': [_P],
    'sgg_act_bwd': [_P, _P, _P, _L, _F, _I, _I, _P],
    'sgg_ce_fwd_bwd': [_P, _I, _P, _I, _I, _I, _P, _F, _F, _P, _I, _P, _I, _P, _P, _I, _I, _F, _F, _P],
    'sgg_label_counts': [_P, _I, _I, _P, _I, _P],
    'sgg_colsum': [_P, _I, _I, _I, _P, _P, _I, _P],
    'sgg_bn_stats': [_P, _I, _I, _P, _P, _I, _P],
    'sgg_bn_finalize': [_P, _I, _I, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P],
    'sgg_bn_apply': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    'sgg_bn_bwd': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _I, _P],
    'sgg_gru_gate_bwd': [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    'sgg_rank4_reduce': [_P, _P, _I, _I, _P, _I, _P, _I, _I, _P],
    'sgg_recall_first_match': [_P, _P, _P, _I, _P, _P, _P, _I, _P, _P, _F, _I, _P, _P, _P],
    'sgg_freq_bias_fwd': [_P, _I, _I, _P, _P, _I, _P, _I, _P, _P, _P, _P, _I, _P],
    'sgg_freq_bias_bwd': [_P, _P, _I, _I, _P, _P],
    'sgg_gemm_tn': [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P],
    'sgg_gemm_tn256': [_P, _I, _P, _I, _P, _I, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P, _P],
    'sgg_sqnorm_multi': [_P, _P, _I, _P, _P, _I, _I, _P],
    'sgg_sgd_multi': [_P, _P, _P, _P, _P, _P, _I, _F, _F, _I, _P, _F, _F, _I, _I, _I, _P, _P],
    'sgg_transpose': [_P, _L, _P, _L, _I, _I, _P, _L, _I, _P, _P, _I, _I, _P],
    'sgg_group_sum': [_P, _L, _P, _L, _I, _I, _I, _I, _P],
    'sgg_pair_slots': [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P],
    'sgg_gemm_addrows': [_P, _I, _P, _I, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    'sgg_transpose_pairsum': [_P, _L, _P, _P, _L, _I, _I, _I, _P],
    'sgg_pairsum': [_P, _L, _P, _P, _L, _I, _I, _I, _P],
    'sgg_group_bcast_add': [_P, _L, _P, _L, _I, _I, _I, _I, _I, _P],
    'sgg_gemm_groupadd': [_P, _I, _P, _I, _P, _I, _I, _I, _P, _I, _I, _I, _I, _I, _I, _P],
    'sgg_add': [_P, _P, _L, _I, _I, _P],
    'sgg_sqnorm_acc': [_P, _L, _P, _P, _I, _I, _P],
    'sgg_sgd_step': [_P, _P, _P, _L, _F, _F, _F, _I, _P, _F, _F, _I, _P],
    'sgg_cast': [_P, _P, _L, _I, _I, _P],
    'sgg_permute_ncp_to_npc': [_P, _P, _I, _I, _I, _I, _I, _P],
}
_RESTYPE = {'sgg_build_info': c_char_p}

_ERRORS = {-1: (ValueError, 'bad argument (size / alignment / null pointer)'),
           -2: (TypeError, 'unsupported element type'),
           -3: (RuntimeError, 'HIP kernel launch failed'),
           -4: (ValueError, 'output capacity too small'),
           -5: (ValueError, 'an operand spans >= 4 GiB (rows are addressed as base + 32-bit lane offset): split the rows over several calls')}

_lib = None


def load():
    """Load (once) and type the shared library.  Raises ImportError if it is absent: build it with
    `python -c "import __graft_entry__ as g; g.build()"` or `make -C sgg_amd/csrc`."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError('sgg_amd: %s not found -- the HIP extension is mandatory (no CPU fallback). '
                          'Build it: make -C sgg_amd/csrc' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    # the version first: a stale build must fail here, before a call with shifted arguments can reach the device
    lib.sgg_abi_version.argtypes, lib.sgg_abi_version.restype = [], c_int
    if lib.sgg_abi_version() != ABI_VERSION:
        raise ImportError('sgg_amd: %s has ABI version %d, this package binds version %d: rebuild (make -C sgg_amd/csrc)'
                          % (LIB_PATH, lib.sgg_abi_version(), ABI_VERSION))
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, c_int)
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        exc, msg = _ERRORS.get(code, (RuntimeError, 'error %d' % code))
        raise exc('%s: %s' % (what, msg))


# Optional per-call timing (bench.py / tests only): when `profiler` is a dict, every C-ABI call is bracketed by
# HIP events recorded on the stream the kernel is launched on; keys are (symbol, tag).
profiler = None
_tag = ['']
_tag_prefix = ['']      # a coarse label in front of the tags the model sets itself (bench.py: 'gan:' around the GAN half of an iteration)


def set_tag(tag):
    _tag[0] = tag


def set_tag_prefix(prefix):
    _tag_prefix[0] = prefix


def call(name, *args):
    if profiler is None:
        check(getattr(load(), name)(*args), name)
        return
    import torch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(getattr(load(), name)(*args), name)
    e1.record()
    profiler.setdefault((name, _tag_prefix[0] + _tag[0]), []).append((e0, e1))
