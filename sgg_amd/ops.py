"""Thin torch-tensor wrappers over the C ABI (include/sgg_hip.h).

torch is used for device memory and the current HIP stream only; every wrapper hands raw device pointers
to libsgg_hip.so.  Nothing here computes on the CPU and nothing falls back to torch ops.
"""
import os

import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, SGG_BF16, SGG_F16, SGG_F32  # noqa: F401

_DT = {torch.float32: SGG_F32, torch.bfloat16: SGG_BF16, torch.float16: SGG_F16}
SGG_PAIR16 = 3      # include/sgg_hip.h: a row / pixel as two f16 planes [hi | lo] (the x3 mode's operand and activation format)


HALF = (torch.bfloat16, torch.float16)     # the two 16-bit storage / MFMA operand formats (same kernels, same rates)


def is_half(t):
    return (t.dtype if isinstance(t, torch.Tensor) else t) in HALF


def _ws(n, device):
    """f32 scratch for a two-stage reduction (partial rows summed in a fixed order: no float atomics anywhere in a step)"""
    return torch.empty(max(int(n), 1), dtype=torch.float32, device=device)


def dt(t):
    try:
        return _DT[t.dtype if isinstance(t, torch.Tensor) else t]
    except KeyError:
        raise TypeError('sgg_amd: unsupported dtype %s (float32 / bfloat16 / float16 only)' % (t,))


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream():
    """Raw handle of torch's current HIP stream on the current device.  The C accessor costs ~0.3 us; going through
    torch.cuda.current_stream() builds a Stream object per call (~8 us x 60-450 calls per forward / train step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _p(t, dtype=None, rows_ok=False):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('sgg_amd: tensor is not on the GPU (the HIP path has no CPU fallback)')
    if not t.is_contiguous() and not (rows_ok and t.dim() == 2 and t.stride(1) == 1):
        raise ValueError('sgg_amd: tensor must be contiguous')
    if dtype is not None and t.dtype != dtype:
        raise TypeError('sgg_amd: expected %s, got %s' % (dtype, t.dtype))
    return t.data_ptr()


def pow2ceil(x):
    p = 1
    while p < x:
        p <<= 1
    return p


# ---------------------------------------------------------------- a-1 / a-2
def image_prep_u8(img_hwc, rh, rw, out_nhwc4, b):
    """img u8[h0,w0,3] (device, as decoded) -> slot b: SquarePad + ToTensor + normalise + resize in one kernel."""
    h0, w0, c = img_hwc.shape
    if c != 3 or img_hwc.dtype != torch.uint8:
        raise TypeError('image_prep_u8: expected a uint8 [h, w, 3] image')
    Hp, Wp = out_nhwc4.shape[1] - 2, out_nhwc4.shape[2] - 2
    _lib.call('sgg_image_prep_u8', _p(img_hwc, torch.uint8), h0, w0, rh, rw, _p(out_nhwc4, torch.float32), b, Hp, Wp,
              _stream())


def image_prep(img_chw, rh, rw, out_nhwc4, b):
    """img f32[3,h,w] (device) -> out f32[B,Hp+2,Wp+2,4] slot b (interior), normalised + resized."""
    _, h, w = img_chw.shape
    Hp, Wp = out_nhwc4.shape[1] - 2, out_nhwc4.shape[2] - 2
    _lib.call('sgg_image_prep', _p(img_chw, torch.float32), h, w, rh, rw, _p(out_nhwc4, torch.float32), b, Hp, Wp,
              _stream())


def image_prep_batch(images, sizes, out_nhwc4):
    """All images of the batch in ONE launch.  images: device tensors, u8 [h,w,3] or f32 [3,h,w]; sizes[k] = (rh, rw) resized size."""
    import numpy as np
    n = len(images)
    keep = [im.contiguous() for im in images]
    for im in keep:
        if not im.is_cuda:
            raise RuntimeError('sgg_amd: tensor is not on the GPU (the HIP path has no CPU fallback)')
    u8 = np.array([1 if im.dtype == torch.uint8 else 0 for im in keep], dtype=np.uint8)
    for im, f in zip(keep, u8):
        if not f and im.dtype != torch.float32:
            raise TypeError('sgg_amd: image must be uint8 [h,w,3] or float32 [3,h,w]')
    ptr = np.array([im.data_ptr() for im in keep], dtype=np.uint64)
    h0 = np.array([im.shape[0] if f else im.shape[-2] for im, f in zip(keep, u8)], dtype=np.int32)
    w0 = np.array([im.shape[1] if f else im.shape[-1] for im, f in zip(keep, u8)], dtype=np.int32)
    rh = np.array([s[0] for s in sizes], dtype=np.int32)
    rw = np.array([s[1] for s in sizes], dtype=np.int32)
    Hp, Wp = out_nhwc4.shape[1] - 2, out_nhwc4.shape[2] - 2
    _lib.call('sgg_image_prep_batch', ptr.ctypes.data, h0.ctypes.data, w0.ctypes.data, rh.ctypes.data, rw.ctypes.data,
              u8.ctypes.data, n, _p(out_nhwc4, torch.float32), Hp, Wp, _stream())


def conv1_1(x_nhwc4, w, bias, out, pair=False):
    """pair=True: `out` is a PAIR plane f16 [B, H+2, W+2, 128] (pixel = [hi (64) | lo (64)]), the x3 mode's form of the fp32 map"""
    B, H, W = x_nhwc4.shape[0], x_nhwc4.shape[1] - 2, x_nhwc4.shape[2] - 2
    assert not pair or (out.dtype == torch.float16 and out.shape[3] == 128)
    _lib.call('sgg_conv1_1', _p(x_nhwc4, torch.float32), _p(w, torch.float32), _p(bias, torch.float32), _p(out),
              B, H, W, SGG_PAIR16 if pair else dt(out), _stream())


def conv1_pack_weights(w1, dtype):
    """conv1_1's weights f32 [64,27] -> the 4-KiB fragment block sgg_conv1_block reads (once per weight change)"""
    frags = torch.empty(4096, dtype=torch.uint8, device=w1.device)
    _lib.call('sgg_conv1_pack_weights', _p(w1, torch.float32), _p(frags), dt(torch.empty(0, dtype=dtype)), _stream())
    return frags


def conv1_block(x_nhwc4, w1_frags, b1, w2, b2, out, out_pad, pool=False):
    """conv1_1 + ReLU + conv1_2 + ReLU (+ MaxPool2d(2)) of VGG-16 in one launch (16-bit `out`); x_nhwc4 f32 [B,H+2,W+2,4];
    w1_frags from conv1_pack_weights (same 16-bit type as `out`)."""
    B, H, W = x_nhwc4.shape[0], x_nhwc4.shape[1] - 2, x_nhwc4.shape[2] - 2
    assert x_nhwc4.dtype == torch.float32 and x_nhwc4.shape[3] == 4 and is_half(out.dtype) and w2.dtype == out.dtype
    assert w1_frags.dtype == torch.uint8 and w1_frags.numel() == 4096
    _lib.call('sgg_conv1_block', _p(x_nhwc4), _p(w1_frags), _p(b1, torch.float32), _p(w2), _p(b2, torch.float32), _p(out), int(out_pad),
              B, H, W, int(bool(pool)), dt(out), _stream())
    return out


def conv_pool_fusable(H, W, Cout):
    """Shapes for which conv3x3_relu(pool=True) exists: the LDS-patch kernel's (wide maps, even sizes)."""
    return H >= 64 and W >= 64 and H % 2 == 0 and W % 2 == 0 and Cout % 64 == 0 and not _SPLIT3[0]     # (x3 mode: conv, then the pool kernel)


def conv3x3_relu(x, w, bias, out, out_pad, pool=False):
    """x [B,H+2,W+2,Cin] zero-bordered; w [Cout,3,3,Cin]; out [B,H+2p,W+2p,Cout], or with pool=True the 2x2-max-pooled
    plane [B,H/2+2p,W/2+2p,Cout] (MaxPool2d(2) fused into the conv epilogue)."""
    B, H, W, Cin = x.shape[0], x.shape[1] - 2, x.shape[2] - 2, x.shape[3]
    Cout = w.shape[0]
    assert w.dtype == x.dtype == out.dtype
    if _SPLIT3[0] and x.dtype == torch.float32 and Cin % 64 == 0:      # (split3 pads K to 64: other widths stay on the exact-fp32 kernel)
        # x3 mode: the plane and the weights as f16 (hi, lo) pairs with 3 Cin channels ([hi | hi | lo] per pixel, [hi | lo | hi] per tap),
        # one f16 implicit-GEMM convolution with fp32 output (a zero border splits into zeros)
        assert not pool
        step = max(1, SPAN_LIMIT // (x.shape[1] * x.shape[2] * 3 * Cin * 2))
        w3 = split3(w.reshape(Cout * 9, Cin), weights=True)
        for b0 in range(0, B, step):
            xb = x[b0:b0 + step]
            x3 = split3(xb.reshape(-1, Cin))
            _lib.call('sgg_conv3x3_relu', _p(x3), _p(w3), _p(bias, torch.float32), _p(out[b0:b0 + step]), out_pad, xb.shape[0], H, W, 3 * Cin, Cout, 0,
                      SGG_F16, SGG_F32, _stream())
        return
    per_image = x.shape[1] * x.shape[2] * Cin * x.element_size()
    if B * per_image > SPAN_LIMIT and B > 1:        # planes of 4 GiB or more (conv1_2 input above ~45 fp32 images): image blocks
        step = max(1, SPAN_LIMIT // per_image)
        for b0 in range(0, B, step):
            conv3x3_relu(x[b0:b0 + step], w, bias, out[b0:b0 + step], out_pad, pool)
        return
    _lib.call('sgg_conv3x3_relu', _p(x), _p(w), _p(bias, torch.float32), _p(out), out_pad, B, H, W, Cin, Cout, int(pool),
              dt(x), dt(out), _stream())


def conv_pp_x3_ok(H, W, Cin, Cout):
    """shapes the patch kernel's x3 form takes (sgg_conv3x3_relu_x3: pair plane in and out, optional fused 2x2 max pool)"""
    return H >= 64 and W >= 64 and Cin % 64 == 0 and Cout % 64 == 0 and os.environ.get('SGG_X3_CONV_PP', '1') != '0'


def conv3x3_relu_x3pp(xp, w, bias, out, out_pad, pool=False):
    """the x3 convolution on the LDS-resident patch kernel (conv_pp.hip): pair plane in, pair plane out (pooled with pool=True)"""
    B, H, W, Cin = xp.shape[0], xp.shape[1] - 2, xp.shape[2] - 2, xp.shape[3] // 2
    Cout = w.shape[0]
    assert xp.dtype == out.dtype == torch.float16 and out.shape[3] == 2 * Cout and conv_pp_x3_ok(H, W, Cin, Cout)
    w3 = split3(w.reshape(Cout * 9, Cin), weights=True)                # [Cout * 9, 3 Cin]: tap = [hi | lo | hi]
    step = max(1, SPAN_LIMIT // (xp.shape[1] * xp.shape[2] * 2 * Cin * 2))
    for b0 in range(0, B, step):
        xb = xp[b0:b0 + step]
        _lib.call('sgg_conv3x3_relu_x3', _p(xb), _p(w3), _p(bias, torch.float32), _p(out[b0:b0 + step]), out_pad, xb.shape[0], H, W, Cin, Cout, int(pool), _stream())


def conv3x3_relu_pair(xp, w, bias, out, out_pad, pair_out):
    """x3 mode on PAIR planes (include/sgg_hip.h SGG_PAIR16): xp f16 [B, H+2, W+2, 2 Cin] zero-bordered (pixel = [hi | lo]), w f32
    [Cout, 3, 3, Cin] (its pair form [Cout, 9, 2 Cin] is made once per weight version and cached); out: a pair plane
    f16 [B, H+2p, W+2p, 2 Cout] (pair_out) or f32 [B, H+2p, W+2p, Cout].  The implicit-GEMM kernels walk hi.hi + hi.lo + lo.hi per tap:
    no split pass over the activations, no [hi | hi | lo] copy of them."""
    B, H, W, Cin = xp.shape[0], xp.shape[1] - 2, xp.shape[2] - 2, xp.shape[3] // 2
    Cout = w.shape[0]
    assert xp.dtype == torch.float16 and Cin % 64 == 0 and w.dtype == torch.float32
    wp = split2(w.reshape(Cout * 9, Cin), weights=True)                # [Cout * 9, 2 Cin]: tap = [hi (Cin) | lo (Cin)]
    assert (out.dtype == torch.float16 and out.shape[3] == 2 * Cout) if pair_out else (out.dtype == torch.float32 and out.shape[3] == Cout)
    step = max(1, SPAN_LIMIT // (xp.shape[1] * xp.shape[2] * 2 * Cin * 2))
    for b0 in range(0, B, step):
        xb = xp[b0:b0 + step]
        _lib.call('sgg_conv3x3_relu', _p(xb), _p(wp), _p(bias, torch.float32), _p(out[b0:b0 + step]), out_pad, xb.shape[0], H, W, Cin, Cout, 0,
                  SGG_PAIR16, SGG_PAIR16 if pair_out else SGG_F32, _stream())


def maxpool2x2_pair(xp, out, out_pad):
    """MaxPool2d(2) on a PAIR plane [B, H+2, W+2, 2 C] -> [B, H/2+2p, W/2+2p, 2 C]"""
    B, H, W, C = xp.shape[0], xp.shape[1] - 2, xp.shape[2] - 2, xp.shape[3] // 2
    assert xp.dtype == out.dtype == torch.float16 and out.shape[3] == 2 * C
    _lib.call('sgg_maxpool2x2', _p(xp), _p(out), out_pad, B, H, W, C, SGG_PAIR16, _stream())


def maxpool2x2(x, out, out_pad):
    B, H, W, C = x.shape[0], x.shape[1] - 2, x.shape[2] - 2, x.shape[3]
    assert x.dtype == out.dtype
    _lib.call('sgg_maxpool2x2', _p(x), _p(out), out_pad, B, H, W, C, dt(x), _stream())


# ---------------------------------------------------------------- a-3
def pair_index_eval(im_inds, boxes=None, require_overlap=False, cap=None):
    """-> (rel_inds i64[cap,3], count int32[1] on device)."""
    N = im_inds.shape[0]
    cap = N * (N - 1) if cap is None else cap
    out = torch.empty((max(cap, 1), 3), dtype=torch.int64, device=im_inds.device)
    count = torch.empty(1, dtype=torch.int32, device=im_inds.device)
    work = torch.empty(N + 2, dtype=torch.int32, device=im_inds.device)
    _lib.call('sgg_pair_index_eval', _p(im_inds, torch.int64), _p(boxes, torch.float32) if require_overlap else None,
              N, int(bool(require_overlap)), _p(out), cap, _p(count), _p(work), _stream())
    return out, count


def pair_index_train(im_inds, gt_rels, img_first, cap):
    N, R = im_inds.shape[0], gt_rels.shape[0]
    out = torch.empty((max(cap, 1), 4), dtype=torch.int64, device=im_inds.device)
    count = torch.empty(1, dtype=torch.int32, device=im_inds.device)
    work = torch.empty(N + 2 + N * N + max(R, 1), dtype=torch.int32, device=im_inds.device)
    _lib.call('sgg_pair_index_train', _p(im_inds, torch.int64), N, _p(gt_rels, torch.int64) if R else None, R,
              _p(img_first, torch.int32) if R else None, _p(out), cap, _p(count), _p(work), _stream())
    return out, count


def rel_assign_tables(det_boxes, det_img, det_labels, gt_boxes, gt_classes, fg_thresh=0.5, filter_non_overlap=True):
    """lib/rel_assignments.py:60-76 for the whole batch -> (gt_iou f32[N,G], match u8[N,G], poss u8[N,N]) on the device."""
    N, G = det_boxes.shape[0], gt_boxes.shape[0]
    dev = det_boxes.device
    gt_iou = torch.empty((N, G), dtype=torch.float32, device=dev)
    match = torch.empty((N, G), dtype=torch.uint8, device=dev)
    poss = torch.empty((N, N), dtype=torch.uint8, device=dev)
    _lib.call('sgg_rel_assign_tables', _p(det_boxes, torch.float32), _p(det_img, torch.int64), _p(det_labels, torch.int64), N,
              _p(gt_boxes, torch.float32) if G else None, _p(gt_classes, torch.int64) if G else None, G, float(fg_thresh),
              int(bool(filter_non_overlap)), _p(gt_iou) if G else None, _p(match) if G else None, _p(poss), _stream())
    return gt_iou, match, poss


def pairs_of(rel_inds):
    """(subject, object) columns of rel_inds i64[E,3] as a contiguous i64[E,2]: the copy a cached edge list carries, or a new one"""
    p = getattr(rel_inds, '_sgg_pairs', None)
    if p is not None and p.shape[0] == rel_inds.shape[0] and p.device == rel_inds.device:
        return p
    return rel_inds[:, 1:].contiguous()


class Csr(tuple):
    """(out_ptr, out_ids, in_ptr, in_ids, so, flags) + what the sliced IMP kernel needs: `img_ptr` i32[B+1] (first node of each
    graph) and `graphs` = (B, max_nodes, max_edges) known on the host; both None when the caller gave no hint."""
    img_ptr = None
    graphs = None


def edge_csr(rel_inds, N, im_inds=None, graphs=None):
    """CSR lists by subject / object.  im_inds (node -> image, i64[N]) may be given when rel_inds is sorted by image.
    graphs = (B, max_nodes, max_edges) (host-side facts, optional): rel_inds is sorted by (image, subject) and no image has
    more nodes / edges than stated -- enables the sliced IMP kernel (imp_sliced_ok)."""
    E = rel_inds.shape[0]
    dev = rel_inds.device
    out_ptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
    in_ptr = torch.empty(N + 1, dtype=torch.int32, device=dev)
    out_ids = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    in_ids = torch.empty(max(E, 1), dtype=torch.int32, device=dev)
    so = torch.empty((max(E, 1), 2), dtype=torch.int32, device=dev)
    flags = torch.empty(1, dtype=torch.int32, device=dev)
    _lib.call('sgg_edge_csr', _p(rel_inds, torch.int64), E, N, _p(im_inds, torch.int64) if im_inds is not None else None,
              _p(out_ptr), _p(out_ids), _p(in_ptr), _p(in_ids), _p(so), _p(flags), _stream())
    csr = Csr((out_ptr, out_ids, in_ptr, in_ids, so, flags))
    if graphs is not None and im_inds is not None and E > 0:
        B = int(graphs[0])
        csr.img_ptr = torch.empty(2 * (B + 1) + 66 * B, dtype=torch.int32, device=dev)   # node offsets, edge offsets, per-graph out-list table
        _lib.call('sgg_graph_ptr', _p(im_inds, torch.int64), N, B, _p(out_ptr), _p(csr.img_ptr), _stream())
        csr.graphs = (B, int(graphs[1]), int(graphs[2]))
    return csr


# ---------------------------------------------------------------- a-4
def roi_align(fmap_nhwc, rois, pairs=None, spatial_scale=1.0 / 16, P=7, sampling=2, add_ec=None, out=None):
    """fmap [B,H,W,C] (NHWC); rois f32[N,5]; pairs i64[R,2] or None -> [R,C,P,P] contiguous (the reference's layout),
    same dtype as fmap."""
    B, H, W, C = fmap_nhwc.shape
    R = rois.shape[0] if pairs is None else pairs.shape[0]
    if out is None:
        out = torch.empty((R, C, P, P), dtype=fmap_nhwc.dtype, device=fmap_nhwc.device)
    _lib.call('sgg_roi_align_fwd', _p(fmap_nhwc), B, H, W, C, _p(rois, torch.float32), rois.shape[0],
              _p(pairs, torch.int64) if pairs is not None else None, R, float(spatial_scale), P, sampling,
              _p(add_ec, torch.float32) if add_ec is not None else None, _p(out), dt(fmap_nhwc), _stream())
    return out


def roi_align_bwd(d_out, fmap_shape, rois, pairs=None, spatial_scale=1.0 / 16, sampling=2, d_fmap=None):
    """d_out [R,C,P,P] -> d_fmap f32 [B,H,W,C] (+= when given): the adjoint of roi_align (GAN path: fmap requires grad).  A gather per
    feature-map cell (no atomics: bit-reproducible); d_out is re-laid out channels-last first so that the gather reads 16-byte pieces."""
    B, H, W, C = fmap_shape
    R, P = d_out.shape[0], d_out.shape[-1]
    if d_fmap is None:
        d_fmap = torch.zeros((B, H, W, C), dtype=torch.float32, device=d_out.device)
    if R == 0:
        return d_fmap
    g = permute_ncp_to_npc(d_out.reshape(R, C, P * P))                   # [R, P*P, C]
    _lib.call('sgg_roi_align_bwd', _p(g), B, H, W, C, _p(rois, torch.float32), rois.shape[0],
              _p(pairs, torch.int64) if pairs is not None else None, R, float(spatial_scale), P, sampling, _p(d_fmap, torch.float32),
              dt(g), 1, _stream())
    return d_fmap


# ---------------------------------------------------------------- a-5 / a-6
def _im_wh(im_sizes, device):
    """[(h, w), ...] per image (the transform's image_sizes) -> f32[B,2] = (w, h) on the device, for the 'raw_boxes' raster"""
    if torch.is_tensor(im_sizes) and im_sizes.is_cuda:
        return im_sizes.float().contiguous()
    return torch.tensor([[float(s[1]), float(s[0])] for s in im_sizes], dtype=torch.float32).to(device, non_blocking=True)


def union_rects(rois, pairs, P=27, offset=-0.5, im_sizes=None):
    """im_sizes None: the 'motifs' raster (boxes inside their union box); given: the 'raw_boxes' raster (image coordinates)."""
    E = pairs.shape[0]
    out = torch.empty((E, 2, P, P), dtype=torch.float32, device=rois.device)
    wh = _im_wh(im_sizes, rois.device) if im_sizes is not None else None
    _lib.call('sgg_union_rects_fwd', _p(rois, torch.float32), _p(pairs, torch.int64), E, P, float(offset), _p(out),
              0 if wh is None else 1, _p(wh), _stream())
    return out


def union_rect_patches(rois, pairs, dtype, P=27, Kpad=128, im_sizes=None):
    E = pairs.shape[0]
    out = torch.empty((E * 4, Kpad), dtype=dtype, device=rois.device)
    wh = _im_wh(im_sizes, rois.device) if im_sizes is not None else None
    _lib.call('sgg_union_rect_patches', _p(rois, torch.float32), _p(pairs, torch.int64), E, P, _p(out), Kpad,
              0 if wh is None else 1, _p(wh), dt(dtype), _stream())
    return out


def max4_rows(x):
    E4, C = x.shape
    out = torch.empty((E4 // 4, C), dtype=x.dtype, device=x.device)
    _lib.call('sgg_max4_rows', _p(x), _p(out), E4 // 4, C, dt(x), _stream())
    return out


def bcast_add_(x, add_rc):
    """x [R,C,PP] += add[R,C] (in place)."""
    R, C, PP = x.shape
    _lib.call('sgg_bcast_add', _p(x), _p(add_rc, torch.float32), R, PP, C, dt(x), _stream())
    return x


# ---------------------------------------------------------------- the x3 mode
# fp32 storage everywhere, every MFMA contraction on f16 SPLIT operands: x = hi + lo (two f16 halves, 22 significand bits), products
# hi.hi + hi.lo + lo.hi accumulated in fp32 by ONE f16 GEMM over the 3 K concatenated segments [hi | hi | lo] . [hi | lo | hi]^T
# (sgg_split3).  The matrix cores' 16-bit rate (2.5 PFLOP/s dense, i.e. 833 TFLOP/s of fp32-grade products) instead of
# v_mfma_f32_32x32x2_f32's 157 TFLOP/s, results within ~1e-6 relative of the exact-fp32 mode: the fast mode that meets the north
# star's 1e-3 clause (RelModelBase.set_compute_dtype(torch.float32, split3=True)).  Process-wide switch: the model sets it.
_SPLIT3 = [False]


def set_split3(on):
    prev = _SPLIT3[0]
    _SPLIT3[0] = bool(on)
    if prev and not on:
        split3_cache_clear()        # leaving the mode: the cached weight splits (up to gigabytes) go back to the allocator
    return prev


def split3_on():
    return _SPLIT3[0]


def _cached_operand(x, tag, make):
    """make(x) for a WEIGHT operand, cached until the tensor changes (same storage, shape, strides and autograd version) -- weights are
    constants in evaluation, and in training they change once per step, not once per contraction."""
    key = None
    if _SPLIT3_CACHE_BYTES > 0:
        key = (tag, x.data_ptr(), x.shape[0], x.shape[1], x.stride(0), x._version, str(x.device))
        hit = _split3_cache.get(key)
        if hit is not None:
            _split3_cache.move_to_end(key)
            cur = _stream()
            if hit[2] != cur:            # made under another stream (the node lane / the main stream): order this stream behind the split
                torch.cuda.current_stream(x.device).wait_event(hit[3])
                hit[1].record_stream(torch.cuda.current_stream(x.device))
            return hit[1]
    out = make(x)
    if key is not None and _split3_seen.get(key, 0) == 0:
        # first sighting of this tensor version: not kept yet -- in training the big weights change every step and are split once per
        # step, caching them would only push a gigabyte of dead splits through the allocator per step (x3 train 21 -> 31 ms when it did);
        # a version that comes back (evaluation; the GRU weights, split several times per step) is kept from its second sighting on
        if len(_split3_seen) > 4096:
            _split3_seen.clear()
        _split3_seen[key] = 1
        key = None
    if key is not None:
        # the entry holds the source tensor too: its storage cannot be recycled for another tensor with the same address while the entry lives.
        # Entries are dropped oldest first beyond SGG_SPLIT3_CACHE_MB (default 6144; 0 = no cache); a split made under one stream is only
        # handed to work issued later on the same device (stream order of the caller, as for every cached operand of model.prepared())
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(x.device))
        _split3_cache[key] = (x, out, _stream(), ev)
        _split3_cache_size[0] += out.numel() * 2
        while _split3_cache_size[0] > _SPLIT3_CACHE_BYTES and len(_split3_cache) > 1:
            _, old = _split3_cache.popitem(last=False)
            _split3_cache_size[0] -= old[1].numel() * 2
    return out


def split3(x, weights=False):
    """x f32 [rows, K] (row-strided ok) -> f16 [rows, 3 * K_pad], K_pad = K rounded up to 64: [hi | hi | lo] (weights: [hi | lo | hi]).
    (K_pad a multiple of 64 keeps 3 K_pad a multiple of the f16 kernels' 64-element K-tile for every K: ADVICE r4.)
    weights=True: the result is cached until the tensor changes (_cached_operand)."""
    rows, K = x.shape
    assert x.dtype == torch.float32 and x.stride(1) == 1

    def make(x):
        Kp = (K + 63) // 64 * 64
        out = torch.empty((rows, 3 * Kp), dtype=torch.float16, device=x.device)
        _lib.call('sgg_split3', _p(x, torch.float32, rows_ok=True), x.stride(0), rows, K, Kp, _p(out), out.stride(0), 1 if weights else 0, _stream())
        return out
    return _cached_operand(x, 'x3', make) if weights else make(x)


def split2(x, weights=False):
    """x f32 [rows, K] (row-strided ok) -> the PAIR form f16 [rows, 2 * K_pad] = [hi | lo] (K_pad = K rounded up to 64): what the x3
    contractions read since round 6 -- the kernels walk (A hi, W hi), (A hi, W lo), (A lo, W hi) over the two planes themselves, so
    nothing is duplicated (4 bytes written per element; sgg_split3's [hi | hi | lo] rows: 6).  weights=True: cached per tensor version."""
    rows, K = x.shape
    assert x.dtype == torch.float32 and x.stride(1) == 1

    def make(x):
        Kp = (K + 63) // 64 * 64
        out = torch.empty((rows, 2 * Kp), dtype=torch.float16, device=x.device)
        _lib.call('sgg_split2', _p(x, torch.float32, rows_ok=True), x.stride(0), rows, K, Kp, _p(out), out.stride(0), _stream())
        return out
    return _cached_operand(x, 'x2', make) if weights else make(x)


PAIR_GEMM = os.environ.get('SGG_X3_PAIR', '1') != '0'      # 0: round 5's [hi | hi | lo] operands (sgg_split3) everywhere


def _gemm_pair(A, W, bias, act, post_scale, post_shift, out, splits):
    """act(A . W^T + bias) * post_scale + post_shift for fp32 A [M,K], W [N,K] on PAIR operands (x3 mode) -> out (fp32 / 16-bit as allocated)"""
    M, K = A.shape
    N = W.shape[0]
    Ap, Wp = split2(A), split2(W, weights=True)
    Kp = Ap.shape[1] // 2
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    kt = 3 * Kp // 64
    can_split = N % 8 == 0 and out.stride(1) == 1 and out.stride(0) % 8 == 0 and out.data_ptr() % 16 == 0
    bias_p = _p(bias, torch.float32) if bias is not None else None
    ps_p = _p(post_scale, torch.float32) if post_scale is not None else None
    pt_p = _p(post_shift, torch.float32) if post_shift is not None else None
    if can_split and ((splits is None and tiles <= 96 and kt >= 64) or (splits is not None and splits > 1)):
        if splits is None:
            splits = max(2, min(16, 512 // tiles, kt // 8))
        splits = min(splits, Kp // 64)
        ws = torch.empty((splits, M, N), dtype=torch.float32, device=A.device)
        _lib.call('sgg_gemm_splitk', _p(Ap), Ap.stride(0), _p(Wp), Wp.stride(0), bias_p, ps_p, pt_p, _p(out, rows_ok=True), out.stride(0), M, N, Kp, act,
                  SGG_PAIR16, dt(out), splits, _p(ws), _stream())
        return out
    _lib.call('sgg_gemm', _p(Ap), Ap.stride(0), None, 0, Kp, _p(Wp), Wp.stride(0), None, 0, bias_p, ps_p, pt_p, _p(out, rows_ok=True), out.stride(0),
              M, N, Kp, act, SGG_PAIR16, dt(out), _stream())
    return out


# x3 mode with an f16 BACKWARD (RelModelBase.set_compute_dtype(torch.float32, split3=True, backward_f16=True)): the forward's contractions
# run on split operands (fp32-grade logits: both parity clauses of the north star are about the forward), the backward's on operands
# rounded to f16 ONCE -- one MFMA product instead of three, under the Trainer's loss scale, i.e. gradients at the accuracy of the f16
# mode (whose 800-step training stays within 0.05 points of the fp32 reference's R@K, tests/test_parity_full_gpu.py).  What standard
# mixed-precision training does, with an fp32-grade forward.  train.PredictFn.backward switches it on for its duration.
_BWD16 = [False]


def set_backward_f16(on):
    prev = _BWD16[0]
    _BWD16[0] = bool(on)
    return prev


def half_operand(x, weights=False):
    """x f32 [rows, K] -> f16 [rows, K_pad] (K_pad = K rounded up to 64, zero columns), rounded once"""
    rows, K = x.shape

    def make(x):
        Kp = (K + 63) // 64 * 64
        if Kp == K:
            return cast(x, torch.float16)
        out = torch.zeros((rows, Kp), dtype=torch.float16, device=x.device)
        out[:, :K].copy_(x)
        return out
    return _cached_operand(x, 'h16', make) if weights else make(x)


import collections  # noqa: E402
_split3_cache = collections.OrderedDict()
_split3_seen = {}
_split3_cache_size = [0]
_SPLIT3_CACHE_BYTES = int(os.environ.get('SGG_SPLIT3_CACHE_MB', '6144')) << 20


def split3_cache_clear():
    _split3_cache.clear()
    _split3_seen.clear()
    _split3_cache_size[0] = 0


def _split3_operands(A, W, A2, W2):
    """the f16 operands [A3 | A23], [W3 | W23] of a split-operand contraction (W given whole [N, K1 + K2] or as W, W2)"""
    K1 = A.shape[1]
    Wa = W if (W2 is not None or A2 is None) else W[:, :K1]
    Wb = W2 if W2 is not None else (W[:, K1:] if A2 is not None else None)
    A3, W3 = split3(A), split3(Wa, weights=True)
    if A2 is None:
        return A3, W3, None, None
    return A3, W3, split3(A2), split3(Wb, weights=True)


# ---------------------------------------------------------------- a-7
def gemm(A, W, bias=None, act=ACT_NONE, out_dtype=None, A2=None, post_scale=None, post_shift=None, out=None, W2=None, splits=None, x3=None):
    """act([A[M,K1] | A2[M,K2]] . [W | W2]^T + bias) * post_scale + post_shift -> [M,N].
    W is [N,K1+K2], or [N,K1] when the second K segment's weights are given separately as W2 [N,K2]."""
    M, K1 = A.shape
    N = W.shape[0]
    K2 = A2.shape[1] if A2 is not None else 0
    K = K1 + K2
    assert A.dtype == W.dtype and (A2 is None or A2.dtype == A.dtype) and (W2 is None or W2.dtype == A.dtype)
    assert W.shape[1] == (K1 if W2 is not None else K), (A.shape, W.shape)
    assert W2 is None or (A2 is not None and tuple(W2.shape) == (N, K2))
    out_dtype = out_dtype or A.dtype
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=A.device)
    if (_SPLIT3[0] if x3 is None else x3) and A.dtype == torch.float32:
        # x3 mode (or a caller that asks for it: x3=True; x3=False keeps an fp32 product exact inside the mode): the same contraction on split
        # f16 operands (3 K columns per segment), fp32 accumulate, `out` as asked
        if _BWD16[0] and x3 is None:
            K1_ = A.shape[1]
            Wa = W if (W2 is not None or A2 is None) else W[:, :K1_]
            Wb = W2 if W2 is not None else (W[:, K1_:] if A2 is not None else None)
            return gemm(half_operand(A), half_operand(Wa, weights=True), bias, act, out_dtype, half_operand(A2) if A2 is not None else None,
                        post_scale, post_shift, out, half_operand(Wb, weights=True) if Wb is not None else None, splits)
        if PAIR_GEMM and A2 is None and W2 is None and M * ((A.shape[1] + 63) // 64 * 64) * 4 < SPAN_LIMIT and A.stride(0) >= A.shape[1]:
            return _gemm_pair(A, W, bias, act, post_scale, post_shift, out, splits)
        A3, W3, A23, W23 = _split3_operands(A, W, A2, W2)
        return gemm(A3, W3, bias, act, out_dtype, A23, post_scale, post_shift, out, W23, splits)
    # the kernels address an operand's rows as (uniform base + 32-bit lane offset): an A operand of 4 GiB or more (fc6 on > 85 k
    # edges in bf16, > 42 k in fp32) goes through in row blocks (SGG_ERR_SPAN is what the entry point returns otherwise)
    row_bytes = max(A.stride(0) * A.element_size(), A2.stride(0) * A2.element_size() if A2 is not None else 0)
    if M * row_bytes > SPAN_LIMIT and M > 1:
        step = max(1, (SPAN_LIMIT // row_bytes) // 256 * 256 or SPAN_LIMIT // row_bytes)
        for m0 in range(0, M, step):
            m1 = min(M, m0 + step)
            gemm(A[m0:m1], W, bias, act, out_dtype, A2[m0:m1] if A2 is not None else None, post_scale, post_shift, out[m0:m1], W2, splits)
        return out
    # short-M, long-K contractions (fc6 on the object rows) do not fill the chip with output tiles: split K
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    kt = K // (64 if is_half(A) else 32)
    can_split = (A2 is None and N % 8 == 0 and out.stride(1) == 1 and out.stride(0) % 8 == 0 and out.data_ptr() % 16 == 0 and
                 A.stride(0) >= K and W.stride(0) >= K)
    if can_split and ((splits is None and tiles <= 96 and kt >= 64) or (splits is not None and splits > 1)):
        if splits is None:
            splits = max(2, min(16, 512 // tiles, kt // 8))
        ws = torch.empty((splits, M, N), dtype=torch.float32, device=A.device)
        _lib.call('sgg_gemm_splitk', _p(A, rows_ok=True), A.stride(0), _p(W, rows_ok=True), W.stride(0),
                  _p(bias, torch.float32) if bias is not None else None,
                  _p(post_scale, torch.float32) if post_scale is not None else None,
                  _p(post_shift, torch.float32) if post_shift is not None else None, _p(out, rows_ok=True), out.stride(0), M, N, K, act, dt(A), dt(out),
                  splits, _p(ws), _stream())
        return out
    _lib.call('sgg_gemm', _p(A, rows_ok=True), A.stride(0), _p(A2, rows_ok=True) if A2 is not None else None,
              A2.stride(0) if A2 is not None else 0, K1, _p(W, rows_ok=True), W.stride(0),
              _p(W2, rows_ok=True) if W2 is not None else None, W2.stride(0) if W2 is not None else 0,
              _p(bias, torch.float32) if bias is not None else None,
              _p(post_scale, torch.float32) if post_scale is not None else None,
              _p(post_shift, torch.float32) if post_shift is not None else None,
              _p(out, rows_ok=True), out.stride(0), M, N, K, act, dt(A), dt(out), _stream())
    return out


def gemm_addrows(A, W, bias, add_rows, add_idx=None, act=ACT_NONE, out_dtype=None, out=None):
    """act(A[M,K] . W[N,K]^T + add_rows[add_idx[m]] + bias) -> [M,N]; add_rows f32 [R,N], add_idx i32 [M] (None: row m)."""
    M, K = A.shape
    N = W.shape[0]
    assert A.dtype == W.dtype and tuple(W.shape) == (N, K) and add_rows.dtype == torch.float32 and add_rows.shape[1] == N
    assert add_idx is None or (add_idx.dtype == torch.int32 and add_idx.numel() == M)
    out_dtype = out_dtype or A.dtype
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=A.device)
    in_dt = None
    if _SPLIT3[0] and A.dtype == torch.float32:
        if _BWD16[0]:
            A, W = half_operand(A), half_operand(W, weights=True)
            K = A.shape[1]
        elif PAIR_GEMM:
            A, W = split2(A), split2(W, weights=True)
            K, in_dt = A.shape[1] // 2, SGG_PAIR16
        else:
            A, W = split3(A), split3(W, weights=True)
            K = A.shape[1]
    _lib.call('sgg_gemm_addrows', _p(A, rows_ok=True), A.stride(0), _p(W, rows_ok=True), W.stride(0),
              _p(bias, torch.float32) if bias is not None else None, _p(add_rows, torch.float32, rows_ok=True), add_rows.stride(0),
              _p(add_idx, torch.int32) if add_idx is not None else None, _p(out, rows_ok=True), out.stride(0), M, N, K, act,
              in_dt if in_dt is not None else dt(A), dt(out), _stream())
    return out


def pair_slots(rel_inds, first, ubase, cnt, U):
    """rel_inds i64[E,3] + per-image tables (i32[B]) -> (e2u i32[E], u2e i32[U,2], flag i32[1]); see include/sgg_hip.h"""
    E, B = rel_inds.shape[0], first.numel()
    dev = rel_inds.device
    e2u = torch.empty(E, dtype=torch.int32, device=dev)
    u2e = torch.empty((U, 2), dtype=torch.int32, device=dev)
    scratch = torch.empty(U + 1, dtype=torch.int32, device=dev)
    _lib.call('sgg_pair_slots', _p(rel_inds, torch.int64), _p(first, torch.int32), _p(ubase, torch.int32), _p(cnt, torch.int32), E, B, U,
              _p(e2u), _p(u2e), scratch.data_ptr(), scratch.data_ptr() + 4 * U, _stream())
    return e2u, u2e, scratch[U:]


def transpose_pairsum(x, u2e, pad_to=64):
    """x [E,C] (bf16 / f32, row-strided ok), u2e i32[U,2] -> [C, Up] in x's dtype: column u = the sum of the pair's edge rows (zero padded)"""
    U, C = u2e.shape[0], x.shape[1]
    Up = (U + pad_to - 1) // pad_to * pad_to
    out = torch.empty((C, Up), dtype=x.dtype, device=x.device)
    _lib.call('sgg_transpose_pairsum', _p(x, rows_ok=True), x.stride(0), _p(u2e, torch.int32), _p(out), Up, U, C, dt(x), _stream())
    return out


def group_bcast_add_(y, r, group, col0=0):
    """y[m, j] += r[m, (j + col0) // group]   (y bf16 / f32 [M, ncol], r f32 [M, C])"""
    M, C = r.shape
    assert y.shape[0] == M and y.shape[1] + col0 <= C * group and r.dtype == torch.float32
    _lib.call('sgg_group_bcast_add', _p(y, rows_ok=True), y.stride(0), _p(r, torch.float32, rows_ok=True), r.stride(0), M, y.shape[1], group,
              col0, dt(y), _stream())
    return y


N_CU = 256     # MI355X
SPAN_LIMIT = 0xffff0000     # bytes one GEMM / conv operand may span (32-bit lane offsets, gemm.hip)


def gemm_full_waves(A, W, out_dtype=None, gadd=None):
    """A[M,K] . W[N,K]^T for the large weight-gradient contractions.  The ping-pong kernel runs one 256x256 tile per CU, so a tile
    count just above a multiple of 256 costs a whole extra round for a few tiles (fc6 weight gradient: 16 x 98 = 1568 tiles = 6.125
    rounds).  When the last round would be at most a quarter full and consists of whole tile columns, those columns are computed by a
    split-K launch of the 128x128 kernel that fills the chip instead, and the main launch is an exact number of rounds.
    gadd = (r f32 [M, N / group], group): out[m, n] += r[m, n // group], in the GEMM's epilogue."""
    M, K = A.shape
    N = W.shape[0]
    tm, tn = M // 256, N // 256
    rem = (tm * tn) % N_CU
    kt = K // (64 if is_half(A) else 32)

    def main(Wp, out, col0=0):
        if gadd is None:
            return gemm(A, Wp, out=out, out_dtype=out_dtype)
        r, group = gadd
        if _SPLIT3[0] and A.dtype == torch.float32:          # x3 mode: the split-operand GEMM, then the group addend as its own pass
            return group_bcast_add_(gemm(A, Wp, out=out, out_dtype=out_dtype), r, group, col0=col0)
        if out is None:
            out = torch.empty((M, Wp.shape[0]), dtype=out_dtype or A.dtype, device=A.device)
        _lib.call('sgg_gemm_groupadd', _p(A, rows_ok=True), A.stride(0), _p(Wp, rows_ok=True), Wp.stride(0), _p(r, torch.float32, rows_ok=True),
                  r.stride(0), group, col0, _p(out, rows_ok=True), out.stride(0), M, Wp.shape[0], K, dt(A), dt(out), _stream())
        return out
    if M % 256 or N % 256 or tm * tn < 2 * N_CU or rem == 0 or rem > N_CU // 4 or rem % tm or kt < 32:   # short reductions: nothing to split
        return main(W, None)
    n1 = N - (rem // tm) * 256
    out = torch.empty((M, N), dtype=out_dtype or A.dtype, device=A.device)
    main(W[:n1], out[:, :n1])
    tail_tiles = (M // 128) * ((N - n1) // 128)
    tiles256 = (M // 256) * ((N - n1) // 256)
    if tiles256 and N_CU // tiles256 >= 2:      # the 256x256 ping-pong kernel's own split-K form: one K slice per CU (32 tiles x 8 slices)
        splits = max(2, min(16, N_CU // tiles256, kt // 8))
    else:
        splits = max(2, min(8, (2 * N_CU) // tail_tiles, kt // 8))
    tail = gemm(A, W[n1:], out=out[:, n1:], splits=splits)     # straight into its columns
    if gadd is not None:
        group_bcast_add_(tail, gadd[0], gadd[1], col0=n1)
    return out


TN256 = os.environ.get('SGG_TN256', '1') != '0'     # the ping-pong kernel's TN form for the large weight gradients (0: transposes + NT)


def _tn_operands_ok(A, B):
    return (is_half(A) and B.dtype == A.dtype and A.dim() == 2 and B.dim() == 2 and A.shape[0] == B.shape[0] and A.shape[0] > 0 and
            A.stride(1) == 1 and B.stride(1) == 1 and A.stride(0) % 8 == 0 and B.stride(0) % 8 == 0 and
            A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0)


def gemm_tn256_ok(A, B, min_tiles=128):
    """Shapes sgg_gemm_tn256 takes and is worth taking (whole 256 x 256 output tiles, half a round of them or more; any row count)."""
    return (TN256 and _tn_operands_ok(A, B) and A.shape[1] % 256 == 0 and B.shape[1] % 256 == 0 and
            (A.shape[1] // 256) * (B.shape[1] // 256) >= min_tiles)


def gemm_tn_ok(A, B):
    """Shapes / dtypes the TN kernels take: one 16-bit format, 16-byte aligned rows, and either reduction rows % 64 with both column
    counts % 128 (sgg_gemm_tn) or whole 256 x 256 tiles with any row count (sgg_gemm_tn256)."""
    return (_tn_operands_ok(A, B) and A.shape[0] % 64 == 0 and A.shape[1] % 128 == 0 and B.shape[1] % 128 == 0) or gemm_tn256_ok(A, B)


def gemm_tn256(A, B, out_dtype=torch.float32, out=None, gadd=None, col0=0):
    """A[Mred,N]^T . B[Mred,K] -> [N,K] on the ping-pong kernel's TN form; gadd = (r f32 [N, .], group): out[n, k] += r[n, (k + col0) // group]."""
    Mred, N = A.shape
    K = B.shape[1]
    if out is None:
        out = torch.empty((N, K), dtype=out_dtype, device=A.device)
    pad = torch.empty((32 * (N + K),), dtype=A.dtype, device=A.device) if Mred % 32 else None
    r, group = gadd if gadd is not None else (None, 1)
    _lib.call('sgg_gemm_tn256', _p(A, rows_ok=True), A.stride(0), _p(B, rows_ok=True), B.stride(0),
              _p(r, torch.float32, rows_ok=True) if r is not None else None, r.stride(0) if r is not None else 0, group, col0,
              _p(out, rows_ok=True), out.stride(0), Mred, N, K, dt(A), dt(out), _p(pad) if pad is not None else None, _stream())
    return out


X3_TN_MIN_TILES = int(os.environ.get('SGG_X3_TN_MIN_TILES', '128'))


def gemm_tn_x3_ok(A, B):
    """fp32 operands the x3 mode's TN form takes (sgg_gemm_tn256 on PAIR operands): whole 256 x 256 tiles, whole 32-row K-tiles"""
    return (PAIR_GEMM and TN256 and A.dtype == B.dtype == torch.float32 and A.shape[0] == B.shape[0] and A.shape[0] % 32 == 0 and
            A.shape[1] % 256 == 0 and B.shape[1] % 256 == 0 and A.stride(1) == 1 and B.stride(1) == 1 and
            (A.shape[1] // 256) * (B.shape[1] // 256) >= X3_TN_MIN_TILES and          # (a few output tiles with a long reduction: the split-K NT form fills the chip)
            A.shape[0] * A.shape[1] * 4 < SPAN_LIMIT and B.shape[0] * B.shape[1] * 4 < SPAN_LIMIT and os.environ.get('SGG_X3_TN', '1') != '0')


def gemm_tn_x3(A, B, out_dtype=torch.float32, out=None):
    """A[Mred,N]^T . B[Mred,K] -> [N,K] for fp32 A, B in the x3 arithmetic, both operands AS THEY LIE (no transposed copies): their PAIR forms
    [Mred, hi | lo] through the ping-pong kernel's TN form, which walks hi.hi + hi.lo + lo.hi over the same reduction rows"""
    Mred, N = A.shape
    K = B.shape[1]
    Ap, Bp = split2(A), split2(B)
    assert Ap.shape[1] == 2 * N and Bp.shape[1] == 2 * K
    if out is None:
        out = torch.empty((N, K), dtype=out_dtype, device=A.device)
    _lib.call('sgg_gemm_tn256', _p(Ap), Ap.stride(0), _p(Bp), Bp.stride(0), None, 0, 1, 0, _p(out, rows_ok=True), out.stride(0), Mred, N, K,
              SGG_PAIR16, dt(out), None, _stream())
    return out


def gemm_tn_full_waves(A, B, out_dtype=None, gadd=None):
    """gemm_full_waves for the TN form (fc6's weight gradient: 16 x 98 tiles = 6.125 rounds): whole rounds on sgg_gemm_tn256, the tile
    columns of a nearly empty last round on the 128 x 128 TN kernel's split form (which needs reduction rows % 64: otherwise one launch)."""
    Mred, N = A.shape
    K = B.shape[1]
    tm, tn = N // 256, K // 256
    rem = (tm * tn) % N_CU
    out = torch.empty((N, K), dtype=out_dtype or A.dtype, device=A.device)
    if tm * tn < 2 * N_CU or rem == 0 or rem > N_CU // 4 or rem % tm or Mred % 64 or Mred < 1024:
        return gemm_tn256(A, B, out=out, gadd=gadd)
    n1 = K - (rem // tm) * 256
    gemm_tn256(A, B[:, :n1], out=out[:, :n1], gadd=gadd)
    tail_tiles = (N // 128) * ((K - n1) // 128)
    splits = max(2, min(8, (2 * N_CU) // tail_tiles, Mred // 512))
    tail = gemm_tn(A, B[:, n1:], out=out[:, n1:], splits=splits)
    if gadd is not None:
        group_bcast_add_(tail, gadd[0], gadd[1], col0=n1)
    return out


def pairsum(x, u2e):
    """x [E, C], u2e i32 [U, 2] -> [U, C] (x's dtype): the sum of every unordered pair's (at most two) edge rows, as rows."""
    U, C = u2e.shape[0], x.shape[1]
    out = torch.empty((U, C), dtype=x.dtype, device=x.device)
    _lib.call('sgg_pairsum', _p(x, rows_ok=True), x.stride(0), _p(u2e, torch.int32), _p(out), out.stride(0), U, C, dt(x), _stream())
    return out


def gemm_tn(A, B, out_dtype=torch.float32, out=None, splits=None):
    """A[Mred,N]^T . B[Mred,K] -> [N,K]: the weight gradient dW = dY^T X without transposed copies of dY and X."""
    Mred, N = A.shape
    K = B.shape[1]
    if splits in (None, 1) and gemm_tn256_ok(A, B):
        return gemm_tn256(A, B, out_dtype=out_dtype, out=out)
    if out is None:
        out = torch.empty((N, K), dtype=out_dtype, device=A.device)
    tiles = (N // 128) * (K // 128)
    if splits is None:
        splits = 1
        if tiles < 256:                       # few output tiles, long reduction: split it so that the chip is full
            splits = max(1, min(16, 512 // tiles, Mred // 256))
    ws = torch.empty((splits, N, K), dtype=torch.float32, device=A.device) if splits > 1 else None
    _lib.call('sgg_gemm_tn', _p(A, rows_ok=True), A.stride(0), _p(B, rows_ok=True), B.stride(0), _p(out, rows_ok=True),
              out.stride(0), Mred, N, K, dt(A), dt(out), splits, _p(ws) if ws is not None else None, _stream())
    return out


# ---------------------------------------------------------------- a-8 / a-9
def imp_sliced_ok(csr, H, dtype):
    """True when the every-row-once kernels take these graphs (host-side facts in csr.graphs, see edge_csr)."""
    g = getattr(csr, 'graphs', None)
    if g is None:
        return False
    cap = _lib.load().sgg_imp_sliced_capacity(H, dt(dtype))
    return g[1] <= 64 and g[2] <= cap


def gate_dots_ok(H):
    """The dot-product epilogue of the GRU gate kernels reduces over the H/8 lanes of a row: H/8 a power of two <= 64."""
    h8 = H // 8
    return H % 8 == 0 and 0 < h8 <= 64 and (h8 & (h8 - 1)) == 0


def imp_ctx(x, csr, N, node_dots, edge_dots, gate_b, pair=2, ctx2=None, ctx_sum=None):
    """The read stream of a message-passing step (sgg_imp_ctx_fwd): ctx2 [2,N,H] = (sum over out-edges g_a x, sum over in-edges g_b x),
    or their sum in ctx_sum [N,H].  pair 2: (out_edge, in_edge) gates on x = e_i -> the context of rel_model_stanford.py:86-91;
    pair 0: (sub_vert, obj_vert) gates on x = d_gi -> the gradient of the node projection.  node_dots f32[N,4] / edge_dots f32[E,4] come
    from the GRU gate kernels (dot_w=...).  Any edge list; graphs promised by edge_csr(graphs=...) go through the every-row-once kernels."""
    E, H = x.shape
    out_ptr, out_ids, in_ptr, in_ids, so, flags = csr
    if ctx_sum is None and ctx2 is None:
        ctx2 = torch.empty((2, N, H), dtype=x.dtype, device=x.device)
    dst = ctx_sum if ctx_sum is not None else ctx2
    g = csr.graphs if imp_sliced_ok(csr, H, x.dtype) else None
    B, max_nodes, max_edges = g if g is not None else (0, 0, 0)
    _lib.call('sgg_imp_ctx_fwd', _p(x), _p(so), _p(out_ptr), _p(out_ids), _p(in_ptr), _p(in_ids),
              _p(csr.img_ptr) if g is not None else None, B, N, E, H, _p(node_dots, torch.float32), _p(edge_dots, torch.float32),
              _p(gate_b, torch.float32), pair, _p(dst), max_edges, max_nodes, 1 if ctx_sum is not None else 0, dt(x), _stream())
    return dst


def gh_dtype(state_dtype):
    """Element type of the edge GRU's hidden pre-activations gh = e W_hh^T + b_hh between the GEMM that writes them and the gate kernel
    that reads them straight back: the state's own 16-bit type (half the bytes of the gate kernel's largest stream: 48.8 MB -> 24.4 MB
    per iteration at 8 images; the added rounding, 2^-11 of a pre-activation, sits inside the 16-bit modes' parity bounds --
    tests/test_parity_full_gpu.py), f32 in the f32 modes.  SGG_GH_F32=1 keeps round 3's f32."""
    return state_dtype if (is_half(state_dtype) and os.environ.get('SGG_GH_F32', '0') != '1') else torch.float32


def gru_gate_proj(gh, P, b_ih, csr, node_dots, edge_dots, gate_b, h_prev, out=None, dot_w=None, dots=None):
    """The edge GRU of a message-passing iteration from the node projection P = v W_ih^T (f32 [N,3H], no bias) instead of e_in rows:
    gi[e] = g_sub P[s] + g_obj P[o] + b_ih (sgg_gru_gate_proj_fwd).  -> h_out (and dots f32[M,4] with dot_w, as gru_gate)."""
    M, H3 = gh.shape
    H = H3 // 3
    so = csr[4]
    if out is None:
        out = torch.empty((M, H), dtype=h_prev.dtype, device=gh.device)
    if dot_w is not None:
        if dot_w.dtype != torch.float32 or dot_w.stride(1) != 1 or dot_w.shape != (4, H):
            raise ValueError('gru_gate_proj: dot_w must be an f32 [4,H] view with unit column stride')
        if dots is None:
            dots = torch.empty((M, 4), dtype=torch.float32, device=gh.device)
    _lib.call('sgg_gru_gate_proj_fwd', _p(gh), _p(P, torch.float32), _p(b_ih, torch.float32), _p(so),
              _p(node_dots, torch.float32), _p(edge_dots, torch.float32), _p(gate_b, torch.float32), _p(h_prev), _p(out), M, H,
              dot_w.data_ptr() if dot_w is not None else None, dot_w.stride(0) if dot_w is not None else 0,
              _p(dots, torch.float32) if dot_w is not None else None, dt(out), dt(gh), _stream())
    return (out, dots) if dot_w is not None else out


def gru_gate(gi, gh, b_hh, h_prev, out_dtype, out=None, dot_w=None, dots=None):
    """dot_w (optional): f32 view [4,H] of the gate matrix [4,2H] (row stride 2H) -- vertex halves gate_w[:, :H] or edge halves
    gate_w[:, H:]; with it the call also returns dots f32[M,4] = h' . dot_w^T (the next IMP step's gate pre-activations)."""
    M, H3 = gi.shape
    H = H3 // 3
    if out is None:
        out = torch.empty((M, H), dtype=out_dtype, device=gi.device)
    if dot_w is not None:
        if dot_w.dtype != torch.float32 or dot_w.stride(1) != 1 or dot_w.shape != (4, H):
            raise ValueError('gru_gate: dot_w must be an f32 [4,H] view with unit column stride')
        if dots is None:
            dots = torch.empty((M, 4), dtype=torch.float32, device=gi.device)
    _lib.call('sgg_gru_gate_fwd', _p(gi), _p(gh) if gh is not None else None,
              _p(b_hh, torch.float32) if b_hh is not None else None, _p(h_prev) if h_prev is not None else None, _p(out),
              M, H, dot_w.data_ptr() if dot_w is not None else None, dot_w.stride(0) if dot_w is not None else 0,
              _p(dots, torch.float32) if dot_w is not None else None, dt(gi), dt(out), _stream())
    return (out, dots) if dot_w is not None else out


# ---------------------------------------------------------------- a-11
def eval_tail(obj_dists, rel_dists, rel_inds, gt_classes=None):
    N, C = obj_dists.shape
    E, P = rel_dists.shape
    dev = obj_dists.device
    obj_scores = torch.empty(N, dtype=torch.float32, device=dev)
    obj_preds = torch.empty(N, dtype=torch.int64, device=dev)
    rels = torch.empty((E, 2), dtype=torch.int64, device=dev)
    pred_scores = torch.empty((E, P), dtype=torch.float32, device=dev)
    work = torch.empty(2 * pow2ceil(max(E, 1)) + E * P + 2, dtype=torch.float32, device=dev)
    _lib.call('sgg_eval_tail', _p(obj_dists), N, C, _p(rel_dists), E, P, _p(rel_inds, torch.int64),
              _p(gt_classes, torch.int64) if gt_classes is not None else None, _p(obj_scores), _p(obj_preds), _p(rels),
              _p(pred_scores), _p(work), dt(obj_dists), _stream())
    return obj_scores, obj_preds, rels, pred_scores


# ---------------------------------------------------------------- utilities
def cast(x, dtype, out=None):
    x = x.contiguous()
    if out is None:
        out = torch.empty(x.shape, dtype=dtype, device=x.device)
    _lib.call('sgg_cast', _p(x), _p(out), x.numel(), dt(x), dt(out), _stream())
    return out


def permute_ncp_to_npc(x, dtype=None):
    """x [N,C,P] -> [N,P,C] (optionally cast)."""
    x = x.contiguous()
    Nn, C, Pp = x.shape
    out = torch.empty((Nn, Pp, C), dtype=dtype or x.dtype, device=x.device)
    if Nn <= 65535:
        _lib.call('sgg_permute_ncp_to_npc', _p(x), _p(out), Nn, C, Pp, dt(x), dt(out), _stream())
    else:
        for s in range(0, Nn, 32768):
            e = min(Nn, s + 32768)
            _lib.call('sgg_permute_ncp_to_npc', x[s:e].data_ptr(), out[s:e].data_ptr(), e - s, C, Pp, dt(x), dt(out),
                      _stream())
    return out


# ---------------------------------------------------------------- training side
def dropout_(x, p, seed, salt=0):
    """nn.Dropout(p) in place.  seed: an int -- the mask of hash(seed * 4 + salt, element) -- or a u64 / i64 [1] DEVICE tensor holding
    that int (read by the kernel at run time: the form a captured hipGraph replays with a new seed, sgg_amd/graph_step.py); same mask."""
    if torch.is_tensor(seed):
        assert seed.is_cuda and seed.numel() == 1 and seed.element_size() == 8
        _lib.call('sgg_dropout_fwd_dev', _p(x), x.numel(), float(p), seed.data_ptr(), int(salt), dt(x), _stream())
    else:
        _lib.call('sgg_dropout_fwd', _p(x), x.numel(), float(p), (int(seed) * 4 + int(salt)) & 0xFFFFFFFFFFFFFFFF, dt(x), _stream())
    return x


def act_bwd(dy, y, scale=1.0):
    dx = torch.empty_like(dy)
    _lib.call('sgg_act_bwd', _p(dy), _p(y), _p(dx), dy.numel(), float(scale), dt(dy), dt(y), _stream())
    return dx


CE_MODES = {'baseline': 0, 'dnorm': 1, 'dnorm-fgbg': 2}


def ce_fwd_bwd(logits, labels, norm, weight, loss, grad, grad_scale=1.0, flag=None, accumulate=True, mode='baseline', alpha=1.0, beta=1.0):
    """loss[0] += (accumulate=False: =) weight / norm[0] * sum CE(logits, labels); grad [M, ldg] (16-bit / f32, zero-padded columns) = grad_scale * d loss / d logits.
    logits f32 [M,C]; labels i64, any 1-D view (e.g. a column of rel_labels); norm, loss: f32 device scalars; flag (optional i32[1]):
    bit 0 is raised when a label lies outside [0, C) (that row adds no loss and gets a zero gradient).
    mode 'dnorm' / 'dnorm-fgbg' (lib/losses.py:44-63): norm = f32[2] (M_FG, M_BG) on the device (label_counts), rows weighted per the
    reference's edge_weights with alpha / beta; `weight` is gamma."""
    M, C = logits.shape
    assert logits.dtype == torch.float32 and logits.stride(1) == 1 and labels.dtype == torch.int64 and labels.dim() == 1
    assert grad.shape[0] == M and grad.shape[1] >= C and grad.stride(1) == 1
    assert norm.dtype == torch.float32 and norm.numel() >= (1 if mode == 'baseline' else 2)
    ws = _ws((M + 3) // 4, logits.device)
    _lib.call('sgg_ce_fwd_bwd', logits.data_ptr(), logits.stride(0), labels.data_ptr(), labels.stride(0), M, C, norm.data_ptr(),
              float(weight), float(grad_scale), loss.data_ptr(), int(bool(accumulate)), grad.data_ptr(), grad.stride(0), _p(ws), _p(flag, torch.int32) if flag is not None else None,
              dt(grad), CE_MODES[mode], float(alpha), float(beta), _stream())


def label_counts(labels, out=None, accumulate=False):
    """f32[2] = (#labels > 0, #labels == 0) on the device: M_FG, M_BG of lib/losses.py:29-34 without a host round trip"""
    assert labels.dtype == torch.int64 and labels.dim() == 1
    if out is None:
        out = torch.empty(2, dtype=torch.float32, device=labels.device)
    _lib.call('sgg_label_counts', labels.data_ptr() if labels.numel() else None, max(labels.stride(0), 1), labels.shape[0], _p(out),
              int(bool(accumulate)), _stream())
    return out


def colsum(x, pool=None):
    """column sums of x[M,N] (bias gradients), f32[N]: row blocks' partial sums added in a fixed order (`pool`: unused, kept for callers)"""
    M, N = x.shape
    out = torch.empty(N, dtype=torch.float32, device=x.device)
    ws = _ws(64 * N, x.device) if M > 512 else None
    _lib.call('sgg_colsum', _p(x, rows_ok=True), M, N, x.stride(0), _p(out), _p(ws) if ws is not None else None, dt(x), _stream())
    return out


def bn_train(x, gamma, beta, run_mean, run_var, eps, momentum, max4, reduce_fn=None):
    """train-mode BatchNorm over the rows of x[M,C] (+ optional max over 4 consecutive rows).
    reduce_fn (data-parallel, synchronised statistics): in-place SUM all-reduce of a small f32 tensor; the batch
    statistics then cover the rows of every rank, as in a single-process step on the concatenated batch.
    -> (y, arg or None, mean, invstd)"""
    M, C = x.shape
    dev = x.device
    sums = torch.empty(2 * C + 1, dtype=torch.float32, device=dev)      # [sum x | sum x^2 | row count]
    _lib.call('sgg_bn_stats', _p(x), M, C, _p(sums), _p(_ws(64 * 2 * C, dev)), dt(x), _stream())
    count_dev = None
    if reduce_fn is not None:
        sums[2 * C:].fill_(float(M))
        reduce_fn(sums)
        count_dev = sums.data_ptr() + 8 * C
    mean, invstd, sc, sh = (torch.empty(C, dtype=torch.float32, device=dev) for _ in range(4))
    _lib.call('sgg_bn_finalize', _p(sums), C, M, count_dev, _p(gamma, torch.float32), _p(beta, torch.float32), float(eps),
              float(momentum), _p(run_mean, torch.float32), _p(run_var, torch.float32), _p(mean), _p(invstd), _p(sc),
              _p(sh), _stream())
    rows_out = M // 4 if max4 else M
    y = torch.empty((rows_out, C), dtype=x.dtype, device=dev)
    arg = torch.empty((rows_out, C), dtype=torch.uint8, device=dev) if max4 else None
    _lib.call('sgg_bn_apply', _p(x), _p(sc), _p(sh), _p(y), _p(arg) if max4 else None, rows_out, C, int(max4), dt(x),
              _stream())
    return y, arg, mean, invstd


def bn_bwd(dy, arg, x, mean, invstd, gamma, max4, reduce_fn=None):
    """-> (dx [rows,C] at the conv output (ReLU folded), dbeta [C], dgamma [C]).  With reduce_fn the two batch sums of
    the backward are synchronised like the forward statistics; dbeta / dgamma stay LOCAL sums (the gradient all-reduce
    adds them up later)."""
    rows, C = x.shape
    dx = torch.empty_like(x)
    sums = torch.empty(2 * C + 1, dtype=torch.float32, device=x.device)
    ws = _ws(64 * 2 * C, x.device)
    args = (_p(dy), _p(arg) if max4 else None, _p(x), _p(mean), _p(invstd), _p(gamma, torch.float32), _p(dx), _p(sums), _p(ws),
            rows, C, int(max4))
    if reduce_fn is None:
        _lib.call('sgg_bn_bwd', *args, 0, None, dt(x), _stream())
        return dx, sums[:C], sums[C:2 * C]
    _lib.call('sgg_bn_bwd', *args, 1, None, dt(x), _stream())
    local = sums[:2 * C].clone()
    sums[2 * C:].fill_(float(rows))
    reduce_fn(sums)
    _lib.call('sgg_bn_bwd', *args, 2, sums.data_ptr() + 8 * C, dt(x), _stream())
    return dx, local[:C], local[C:]


def gru_gate_bwd(dh, gi, gh, b_hh, h_prev, d_gi, d_gh, want_dh_prev=True):
    M, H = dh.shape
    dh_prev = torch.empty_like(dh) if want_dh_prev else None
    _lib.call('sgg_gru_gate_bwd', _p(dh), _p(gi, torch.float32), _p(gh, torch.float32) if gh is not None else None,
              _p(b_hh, torch.float32) if b_hh is not None else None, _p(h_prev) if h_prev is not None else None,
              _p(d_gi), _p(d_gh), _p(dh_prev) if want_dh_prev else None, M, H, dt(dh), _stream())
    return dh_prev


def gru_gate_proj_bwd(dh, gh, P, b_ih, csr, node_dots, edge_dots, gate_b, h_prev, d_gi, d_gh):
    """-> (dh_prev, dq f32[M,2]); d_gi / d_gh [M,3H] are written (sgg_gru_gate_proj_bwd)."""
    M, H = dh.shape
    dh_prev = torch.empty_like(dh)
    dq = torch.empty((M, 2), dtype=torch.float32, device=dh.device)
    _lib.call('sgg_gru_gate_proj_bwd', _p(dh), _p(gh), _p(P, torch.float32), _p(b_ih, torch.float32), _p(csr[4]),
              _p(node_dots, torch.float32), _p(edge_dots, torch.float32), _p(gate_b, torch.float32), _p(h_prev), _p(d_gi), _p(d_gh),
              _p(dh_prev), _p(dq), M, H, dt(dh), dt(gh), _stream())
    return dh_prev, dq


def imp_edge_ctx_bwd(e, csr, node_dots, edge_dots, gate_w, gate_b, dq, d_ctx, d_e, da=None):
    """d_e += the step's gate / context terms; -> da f32[E,4] (sgg_imp_edge_ctx_bwd)."""
    E, H = e.shape
    if da is None:
        da = torch.empty((E, 4), dtype=torch.float32, device=e.device)
    _lib.call('sgg_imp_edge_ctx_bwd', _p(e), _p(csr[4]), E, H, _p(node_dots, torch.float32), _p(edge_dots, torch.float32),
              _p(gate_w, torch.float32), _p(gate_b, torch.float32), _p(dq, torch.float32), _p(d_ctx), _p(d_e), _p(da), dt(e), _stream())
    return da


def imp_node_gates_bwd(da, csr, gate_w, d_v, nsum=None):
    """d_v += sum_k S_k w_k[:H]; -> nsum f32[N,4] (sgg_imp_node_gates_bwd)."""
    N, H = d_v.shape
    out_ptr, out_ids, in_ptr, in_ids = csr[:4]
    if nsum is None:
        nsum = torch.empty((N, 4), dtype=torch.float32, device=d_v.device)
    _lib.call('sgg_imp_node_gates_bwd', _p(da, torch.float32), _p(out_ptr), _p(out_ids), _p(in_ptr), _p(in_ids), _p(gate_w, torch.float32),
              N, H, _p(d_v), _p(nsum), dt(d_v), _stream())
    return nsum


def rank4_reduce_(a, x, out, col0=0, accumulate=True):
    """out[k, col0:col0+H] += (accumulate=False: =) sum_r a[r,k] * x[r,:]   (out f32 [4, ld])"""
    R, H = x.shape
    _lib.call('sgg_rank4_reduce', _p(a, torch.float32), _p(x), R, H, out.data_ptr() + 4 * col0, out.stride(0), _p(_ws(64 * 4 * H, x.device)),
              int(bool(accumulate)), dt(x), _stream())


def transpose(x, pad_to=64, dtype=None, add=None, group=1, want_colsum=False, out=None):
    """x [R,C] (row-strided ok) -> [C, Rp] with Rp = R rounded up to `pad_to`, zero padded.
    add f32[R, C/group]: out[c][r] = x[r][c] + add[r][c // group].  want_colsum: also return the column sums of x (f32[C]).
    out (optional): a [C, Rp] buffer of an earlier call with the same shapes -- its padding columns are zero already and stay so."""
    R, C = x.shape
    Rp = (R + pad_to - 1) // pad_to * pad_to
    dtype = dtype or x.dtype
    if out is None:
        out = (torch.zeros if Rp != R else torch.empty)((C, Rp), dtype=dtype, device=x.device)
    else:
        assert tuple(out.shape) == (C, Rp) and out.dtype == dtype and out.is_contiguous()
    cs = torch.empty(C, dtype=torch.float32, device=x.device) if want_colsum else None
    ws = _ws((R + 63) // 64 * C, x.device) if want_colsum else None
    _lib.call('sgg_transpose', _p(x, rows_ok=True), x.stride(0), _p(out), Rp, R, C,
              _p(add, torch.float32) if add is not None else None, add.stride(0) if add is not None else 0, group,
              _p(cs) if want_colsum else None, _p(ws) if want_colsum else None, dt(x), dt(out), _stream())
    return (out, cs) if want_colsum else out


def transpose_multi(pairs):
    """[(x [R,C] 16-bit (row-strided ok), out [C, Rp >= R])] -> every out = x^T in ONE launch (sgg_transpose_multi; at most 16 per launch).  The outputs'
    padding columns are left as they are (zero since their allocation)."""
    import ctypes
    for i0 in range(0, len(pairs), 16):
        chunk = pairs[i0:i0 + 16]
        n = len(chunk)
        dtc = dt(chunk[0][0])
        PA, LA, IA = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n
        for x, o in chunk:
            assert x.dim() == 2 and x.stride(1) == 1 and o.is_contiguous() and o.shape[0] == x.shape[1] and o.shape[1] >= x.shape[0] and dt(x) == dt(o) == dtc
        _lib.call('sgg_transpose_multi', PA(*[x.data_ptr() for x, _ in chunk]), LA(*[x.stride(0) for x, _ in chunk]), PA(*[o.data_ptr() for _, o in chunk]),
                  LA(*[o.stride(0) for _, o in chunk]), IA(*[x.shape[0] for x, _ in chunk]), IA(*[x.shape[1] for x, _ in chunk]), n, dtc, _stream())


def group_sum(w, C, group, dtype, out=None):
    """w f32[N, C*group] -> [N, C] in dtype: sums over each run of `group` consecutive columns (out: an earlier result's buffer, rewritten)."""
    Nn = w.shape[0]
    if out is None or tuple(out.shape) != (Nn, C) or out.dtype != dtype or out.device != w.device or not out.is_contiguous():
        out = torch.empty((Nn, C), dtype=dtype, device=w.device)
    _lib.call('sgg_group_sum', _p(w, torch.float32), w.stride(0), _p(out), C, Nn, C, group, dt(out), _stream())
    return out


def add_(y, x):
    _lib.call('sgg_add', _p(y), _p(x), y.numel(), dt(y), dt(x), _stream())
    return y


# ---------------------------------------------------------------- ResNet-50-FPN glue (GQA configuration)
def im2col(x, k, stride, pad, src_pad=0, C=None, Kp=None, dtype=None):
    """Patch matrix of a k x k / stride / pad convolution: x [B, H+2 src_pad, W+2 src_pad, Ca] -> ([B*Ho*Wo, Kp], Ho, Wo); columns
    (ky, kx, c) over the first C channels, zero-filled to Kp (default: the next multiple of 64)."""
    B, H, W, Ca = x.shape[0], x.shape[1] - 2 * src_pad, x.shape[2] - 2 * src_pad, x.shape[3]
    C = Ca if C is None else C
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    Kp = (k * k * C + 63) // 64 * 64 if Kp is None else Kp
    out = torch.empty((B * Ho * Wo, Kp), dtype=dtype or x.dtype, device=x.device)
    _lib.call('sgg_im2col', _p(x), B, H, W, Ca, C, src_pad, k, stride, pad, Ho, Wo, _p(out), Kp, dt(x), dt(out), _stream())
    return out, Ho, Wo


def col2im(d_cols, shape, k, stride, pad):
    """Adjoint of im2col on a dense NHWC plane: d_cols [B*Ho*Wo, Kp] -> d_x of `shape` = (B, H, W, C)."""
    B, H, W, C = shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    assert d_cols.shape[0] == B * Ho * Wo and d_cols.is_contiguous()
    out = torch.empty(shape, dtype=d_cols.dtype, device=d_cols.device)
    _lib.call('sgg_col2im', _p(d_cols), B, H, W, C, k, stride, pad, Ho, Wo, d_cols.shape[1], _p(out), dt(out), _stream())
    return out


def maxpool3x3s2(x):
    """MaxPool2d(3, stride=2, padding=1) on NHWC [B,H,W,C]."""
    B, H, W, C = x.shape
    out = torch.empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), dtype=x.dtype, device=x.device)
    _lib.call('sgg_maxpool3x3s2', _p(x), _p(out), B, H, W, C, dt(x), _stream())
    return out


def plane_copy(x, out, src_pad=0, dst_pad=0, stride=1):
    """out[b,y,x,:] = x[b, y*stride, x*stride, :] between NHWC planes with borders src_pad / dst_pad (only interiors are written)."""
    B, C = x.shape[0], x.shape[3]
    assert out.shape[0] == B and out.shape[3] == C and out.dtype == x.dtype
    _lib.call('sgg_plane_copy', _p(x), x.shape[1] - 2 * src_pad, x.shape[2] - 2 * src_pad, src_pad, _p(out), out.shape[1] - 2 * dst_pad,
              out.shape[2] - 2 * dst_pad, dst_pad, B, C, stride, dt(x), _stream())
    return out


def upsample_add_(y, top):
    """y [B,H,W,C] += nearest-neighbour upsampling of top [B,Ht,Wt,C] to (H,W), in place"""
    B, H, W, C = y.shape
    assert top.shape[0] == B and top.shape[3] == C and top.dtype == y.dtype
    _lib.call('sgg_upsample_add', _p(y), _p(top), B, H, W, top.shape[1], top.shape[2], C, dt(y), _stream())
    return y


def add_relu_(y, x):
    """y = max(y + x, 0) in place"""
    assert y.shape == x.shape and y.dtype == x.dtype
    _lib.call('sgg_add_relu', _p(y), _p(x), y.numel(), dt(y), _stream())
    return y
