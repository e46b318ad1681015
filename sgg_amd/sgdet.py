"""SGDet front end after the backbone (SURVEY a-12): what `detector.rpn`, `detector.roi_heads` and
`detector.transform.postprocess` do in eval mode at sgg_models/rel_model_base.py:210-213 ([3P] torchvision FasterRCNN),
on the HIP path.  Dense parts: sgg_conv3x3_relu (RPN 3x3), sgg_gemm (1x1 heads, box head fc6/fc7, predictor),
sgg_roi_align_fwd; the rest in csrc/det.hip.  Eval only -- torchvision's RoIHeads returns no detections in training mode,
which is why the reference documents SGDet training as unsupported (README.md:214-218).
"""
import ctypes
import math

import torch

from . import _lib, ops

ANCHOR_SIZES = (32, 64, 128, 256, 512)      # rel_model_base.py:94
ANCHOR_RATIOS = (0.5, 1.0, 2.0)             # rel_model_base.py:95
RPN_PRE_NMS, RPN_POST_NMS, RPN_NMS_THRESH, RPN_MIN_SIZE = 1000, 1000, 0.7, 1e-3   # [3P] FasterRCNN test-time defaults
DET_MIN_SIZE = 1e-2                          # [3P] RoIHeads.postprocess_detections
DET_PRE_NMS_CAP = 4096                       # documented deviation: candidates entering the per-class NMS


def base_anchors(device):
    """[3P] AnchorGenerator.generate_anchors (ratio-major, scale-minor, rounded); 15 x 4 floats, computed on the host."""
    out = []
    for r in ANCHOR_RATIOS:
        hr = math.sqrt(r)
        wr = 1.0 / hr
        for s in ANCHOR_SIZES:
            w, h = torch.tensor(wr, dtype=torch.float32) * s, torch.tensor(hr, dtype=torch.float32) * s
            out.append([float(torch.round(-w / 2)), float(torch.round(-h / 2)), float(torch.round(w / 2)),
                        float(torch.round(h / 2))])
    return torch.tensor(out, dtype=torch.float32, device=device)


def _sort_desc(keys, seg_off, nseg, seg_len_hint=0):
    """stable descending sort per segment -> (keys_sorted, pos_in_segment)."""
    n = keys.numel()
    dev = keys.device
    keys_out = torch.empty_like(keys)
    vals_tmp = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    vals_out = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    nbytes = ctypes.c_size_t(0)
    stream = ops._stream()
    _lib.call('sgg_segmented_sort_desc', None, None, None, None, n, nseg, None, seg_len_hint, None, ctypes.byref(nbytes), stream)
    temp = torch.empty(max(int(nbytes.value), 8), dtype=torch.uint8, device=dev)
    _lib.call('sgg_segmented_sort_desc', keys.data_ptr(), keys_out.data_ptr(), vals_tmp.data_ptr(), vals_out.data_ptr(), n,
              nseg, seg_off.data_ptr(), seg_len_hint, temp.data_ptr(), ctypes.byref(nbytes), stream)
    return keys_out, vals_out


TOPK_MAX = 4096          # sgg_topk_gather's capacity (one workgroup's LDS)
USE_TOPK = True          # tests flip it to cross-check against the library sort


def _topk_gather(scores, seg, boxes, labels, img_hw, B, take, min_size, pb, ps, pl, pv):
    """radix select + sort of the `take` best candidates per segment (sgg_topk_gather) instead of a full sort of all of them"""
    _lib.call('sgg_topk_gather', scores.data_ptr(), seg.data_ptr(), boxes.data_ptr(), labels.data_ptr() if labels is not None else None,
              img_hw.data_ptr(), B, take, float(min_size), pb.data_ptr(), ps.data_ptr(), pl.data_ptr() if pl is not None else None,
              pv.data_ptr(), ops._stream())


def _nms(boxes, labels, valid, thresh, max_keep):
    B, n = boxes.shape[0], boxes.shape[1]
    dev = boxes.device
    ws = torch.empty(B * n * ((n + 63) // 64), dtype=torch.int64, device=dev)
    keep_idx = torch.empty((B, max_keep), dtype=torch.int32, device=dev)
    keep_cnt = torch.empty(B, dtype=torch.int32, device=dev)
    _lib.call('sgg_nms', boxes.data_ptr(), labels.data_ptr() if labels is not None else None, valid.data_ptr(), B, n,
              float(thresh), max_keep, ws.data_ptr(), keep_idx.data_ptr(), keep_cnt.data_ptr(),
              ops._stream())
    return keep_idx, keep_cnt


def prepared(model, fpn=False):
    """RPN / box-head weights in kernel layout (the detector is frozen: built once)."""
    det = model.detector
    dt = model.compute_dtype
    key = ('sgdet', dt, fpn, det.rpn.head.conv.weight.data_ptr(), det.rpn.head.conv.weight._version)
    if getattr(det, '_sgdet_prep', {}).get('key') == key:
        return det._sgdet_prep['val']
    f = lambda t: t.detach().float().contiguous()
    cast = lambda t: t if t.dtype == dt else ops.cast(t, dt)
    h, rh = det.rpn.head, det.roi_heads
    w = {}
    cw = f(h.conv.weight)
    w['rpn_conv'] = ops.permute_ncp_to_npc(cw.reshape(cw.shape[0], cw.shape[1], 9), dt).view(cw.shape[0], 3, 3, cw.shape[1])
    w['rpn_conv_b'] = f(h.conv.bias)
    A = h.cls_logits.weight.shape[0]
    w['rpn_head'] = cast(torch.cat((f(h.cls_logits.weight).reshape(A, -1), f(h.bbox_pred.weight).reshape(4 * A, -1)), 0))
    w['rpn_head_b'] = torch.cat((f(h.cls_logits.bias), f(h.bbox_pred.bias)))
    w['A'] = A
    w['fc6'], w['fc6_b'] = cast(f(rh.box_head.fc6.weight)), f(rh.box_head.fc6.bias)
    w['fc7'], w['fc7_b'] = cast(f(rh.box_head.fc7.weight)), f(rh.box_head.fc7.bias)
    w['pred'] = cast(torch.cat((f(rh.box_predictor.cls_score.weight), f(rh.box_predictor.bbox_pred.weight)), 0))
    w['pred_b'] = torch.cat((f(rh.box_predictor.cls_score.bias), f(rh.box_predictor.bbox_pred.bias)))
    w['C'] = rh.box_predictor.cls_score.weight.shape[0]
    w['anchors'] = [level_anchors(sz, cw.device) for sz in FPN_ANCHOR_SIZES] if fpn else [base_anchors(cw.device)]
    if w['anchors'][0].shape[0] != A:
        raise ValueError('the RPN head predicts %d anchors per location, the anchor table has %d' % (A, w['anchors'][0].shape[0]))
    det._sgdet_prep = dict(key=key, val=w)
    return w


FPN_ANCHOR_SIZES = (32, 64, 128, 256, 512)  # [3P] maskrcnn_resnet50_fpn: one anchor size per pyramid level (P2 .. P5, pool), 3 ratios each
FPN_CANONICAL = (224.0, 4)                   # [3P] LevelMapper: canonical box size / level (k0), eps 1e-6


def level_anchors(size, device):
    """[3P] AnchorGenerator.generate_anchors for ONE size: 3 x 4 floats (ratio order), rounded"""
    out = []
    for r in ANCHOR_RATIOS:
        hr = math.sqrt(r)
        w, h = torch.tensor(1.0 / hr, dtype=torch.float32) * size, torch.tensor(hr, dtype=torch.float32) * size
        out.append([float(torch.round(-w / 2)), float(torch.round(-h / 2)), float(torch.round(w / 2)), float(torch.round(h / 2))])
    return torch.tensor(out, dtype=torch.float32, device=device)


def _rpn_level(w, fm, anchors, padded_hw):
    """RPN head on one feature map [B,Hf,Wf,C]: 3x3 conv + ReLU, the two 1x1 convolutions as one GEMM over the pixels, anchor decoding
    -> (boxes f32[B, Hf*Wf*A, 4], objectness logits f32[B, Hf*Wf*A]) in (y, x, anchor) order."""
    B, Hf, Wf, Cf = fm.shape
    dt, dev, A = fm.dtype, fm.device, anchors.shape[0]
    fpad = torch.zeros((B, Hf + 2, Wf + 2, Cf), dtype=dt, device=dev)
    ops.plane_copy(fm.contiguous(), fpad, dst_pad=1)
    t = torch.empty((B, Hf, Wf, Cf), dtype=dt, device=dev)
    ops.conv3x3_relu(fpad, w['rpn_conv'], w['rpn_conv_b'], t, 0)
    head = ops.gemm(t.view(B * Hf * Wf, Cf), w['rpn_head'], w['rpn_head_b'], out_dtype=torch.float32)     # [B*HW, 5A]
    n_anch = Hf * Wf * A
    boxes = torch.empty((B, n_anch, 4), dtype=torch.float32, device=dev)
    scores = torch.empty((B, n_anch), dtype=torch.float32, device=dev)
    _lib.call('sgg_rpn_decode', head.data_ptr(), head.stride(0), anchors.data_ptr(), A, Hf, Wf,
              float(padded_hw[0] // Hf), float(padded_hw[1] // Wf), B, boxes.data_ptr(), scores.data_ptr(), ops._stream())
    return boxes, scores


def _top_per_image(boxes, scores, labels, img_hw, take, min_size):
    """the `take` best-scoring boxes of every image, best first: -> (boxes [B,take,4] clipped to the image, scores, labels or None,
    valid u8 [B,take]: a finite score and both sides >= min_size)"""
    B, n = scores.shape
    dev = scores.device
    seg = torch.arange(0, B + 1, dtype=torch.int32, device=dev) * n
    pb = torch.empty((B, take, 4), dtype=torch.float32, device=dev)
    ps = torch.empty((B, take), dtype=torch.float32, device=dev)
    pl = torch.empty((B, take), dtype=torch.int32, device=dev) if labels is not None else None
    pv = torch.empty((B, take), dtype=torch.uint8, device=dev)
    if take <= TOPK_MAX and USE_TOPK:
        _topk_gather(scores.reshape(-1), seg, boxes, labels, img_hw, B, take, min_size, pb, ps, pl, pv)
        return pb, ps, pl, pv
    ks, vs = _sort_desc(scores.reshape(-1), seg, B, n)
    _lib.call('sgg_gather_topk', ks.data_ptr(), vs.data_ptr(), seg.data_ptr(), boxes.data_ptr(), labels.data_ptr() if labels is not None else None,
              img_hw.data_ptr(), B, take, float(min_size), pb.data_ptr(), ps.data_ptr(), pl.data_ptr() if pl is not None else None,
              pv.data_ptr(), ops._stream())
    return pb, ps, pl, pv


def propose(model, w, maps, img_hw, padded_hw):
    """[3P] RegionProposalNetwork.filter_proposals at test time.  maps: the feature maps the RPN head runs on -- one (VGG) or five
    (P2 .. P5 + pool).  Per map the RPN_PRE_NMS best anchors by objectness; all of an image's candidates, best first, clipped, small
    ones dropped, through NMS 0.7 that only compares boxes of the same map (batched_nms over the levels); the best RPN_POST_NMS kept.
    -> (rois f32[K,5], offsets per image on the host)"""
    dev = maps[0].device
    B = maps[0].shape[0]
    per_level = []
    for li, fm in enumerate(maps):
        boxes, scores = _rpn_level(w, fm, w['anchors'][li], padded_hw)
        if len(maps) == 1:
            per_level.append((boxes, scores, None))
            break
        take = min(RPN_PRE_NMS, scores.shape[1])
        pb, ps, _, _ = _top_per_image(boxes, scores, None, img_hw, take, 0.0)
        per_level.append((pb, ps, torch.full((B, take), li, dtype=torch.int32, device=dev)))
    if len(per_level) == 1:
        boxes, scores, levels = per_level[0]
        take = min(RPN_PRE_NMS, scores.shape[1])
    else:
        boxes = torch.cat([t[0] for t in per_level], 1).contiguous()
        scores = torch.cat([t[1] for t in per_level], 1).contiguous()
        levels = torch.cat([t[2] for t in per_level], 1).contiguous()
        take = scores.shape[1]
    pb, ps, pl, pv = _top_per_image(boxes, scores, levels, img_hw, take, RPN_MIN_SIZE)
    keep_idx, keep_cnt = _nms(pb, pl, pv, RPN_NMS_THRESH, min(RPN_POST_NMS, take))
    rois = torch.empty((B * take, 5), dtype=torch.float32, device=dev)
    offs = torch.empty(B + 1, dtype=torch.int32, device=dev)
    _lib.call('sgg_compact_rois', pb.data_ptr(), keep_idx.data_ptr(), keep_cnt.data_ptr(), B, take, keep_idx.shape[1],
              rois.data_ptr(), offs.data_ptr(), ops._stream())
    offs_h = offs.cpu().tolist()                                 # one D2H read: proposal counts size the box-head GEMMs
    return rois[:offs_h[-1]], offs_h


def pyramid_level_of(rois, n_levels, k_min=2):
    """[3P] LevelMapper (FPN paper eq. 1): level k = floor(k0 + log2(sqrt(area) / 224) + 1e-6) clamped to the pyramid; -> i64[K] in [0, n_levels)"""
    s0, k0 = FPN_CANONICAL
    area = (rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2])
    k = torch.floor(k0 + torch.log2(torch.sqrt(area) / s0) + 1e-6)
    return (k.clamp(k_min, k_min + n_levels - 1) - k_min).to(torch.int64)


def box_features(model, maps, scales, rois):
    """[3P] MultiScaleRoIAlign: every RoI pooled (7x7, 2x2 samples per bin) from the map its size selects -> [K, C*P*P]"""
    if len(maps) == 1:
        return ops.roi_align(maps[0], rois, None, scales[0], model.pool_sz, 2).view(rois.shape[0], -1)
    K = rois.shape[0]
    lv = pyramid_level_of(rois, len(maps))
    order = torch.argsort(lv, stable=True)
    counts = torch.bincount(lv, minlength=len(maps)).cpu().tolist()
    C = maps[0].shape[3]
    sorted_feat = torch.empty((K, C * model.pool_sz ** 2), dtype=maps[0].dtype, device=rois.device)
    s = 0
    for li, n in enumerate(counts):
        if n:
            sub = rois[order[s:s + n]].contiguous()
            sorted_feat[s:s + n] = ops.roi_align(maps[li].contiguous(), sub, None, scales[li], model.pool_sz, 2).view(n, -1)
        s += n
    feat = torch.empty_like(sorted_feat)
    feat[order] = sorted_feat
    return feat


def detect(model, fmap, image_sizes, padded_hw, orig_sizes, spatial_scale, pyramid=None):
    """fmap [B,Hf,Wf,C] NHWC (compute dtype): the map the relation model reads.  pyramid None: it is also the detector's only map
    (VGG); else the list [P2 .. P5] and fmap is the 'pool' level -- the RPN runs on all five, the box head pools from the four
    ([3P] maskrcnn_resnet50_fpn).  Returns per-image lists (boxes in resized space, boxes in original space, labels i64, scores) --
    rel_model_base.py:215-232 consumes them."""
    w = prepared(model, fpn=pyramid is not None)
    det = model.detector
    dev = fmap.device
    B = fmap.shape[0]
    stream = ops._stream()
    C = w['C']
    img_hw = torch.tensor([[float(s[0]), float(s[1])] for s in image_sizes], dtype=torch.float32, device=dev)
    if pyramid is None:
        rpn_maps, box_maps, scales = [fmap], [fmap], [spatial_scale]
    else:
        rpn_maps, box_maps = list(pyramid) + [fmap], list(pyramid)
        top = float(max(s[0] for s in image_sizes))
        scales = [2.0 ** round(math.log2(float(m.shape[1]) / top)) for m in box_maps]      # [3P] MultiScaleRoIAlign.infer_scale
    rois, offs_h = propose(model, w, rpn_maps, img_hw, padded_hw)
    K = offs_h[-1]
    det.last_proposals = K          # (bench.py prices the box head's contractions with it)
    det.last_proposal_offsets = list(offs_h)        # per image: [0, k0, k0 + k1, ...] (tests assert the config's 1 000 per image)
    det.last_rois = rois
    if K == 0:
        raise ValueError('at least two objects must be detected to build relationships, make sure the detector is properly '
                         'pretrained', [])
    # ---- RoI heads: RoIAlign -> fc6 -> fc7 -> (cls_score | bbox_pred)
    feat = box_features(model, box_maps, scales, rois)
    tag0 = _lib._tag[0]
    _lib.set_tag('box_fc6')
    x = ops.gemm(feat, w['fc6'], w['fc6_b'], ops.ACT_RELU)
    _lib.set_tag('box_fc7')
    x = ops.gemm(x, w['fc7'], w['fc7_b'], ops.ACT_RELU)
    _lib.set_tag('box_pred')
    pred = ops.gemm(x, w['pred'], w['pred_b'], out_dtype=torch.float32)                                    # [K, 5C]
    _lib.set_tag(tag0)
    ncand = K * (C - 1)
    cs = torch.empty(ncand, dtype=torch.float32, device=dev)
    cb = torch.empty((ncand, 4), dtype=torch.float32, device=dev)
    cl = torch.empty(ncand, dtype=torch.int32, device=dev)
    _lib.call('sgg_det_candidates', pred.data_ptr(), pred.stride(0), rois.data_ptr(), K, C, img_hw.data_ptr(),
              float(det.roi_heads.score_thresh), DET_MIN_SIZE, cs.data_ptr(), cb.data_ptr(), cl.data_ptr(), stream)
    cseg = torch.tensor([o * (C - 1) for o in offs_h], dtype=torch.int32, device=dev)
    cap = DET_PRE_NMS_CAP
    db = torch.empty((B, cap, 4), dtype=torch.float32, device=dev)
    ds = torch.empty((B, cap), dtype=torch.float32, device=dev)
    dl = torch.empty((B, cap), dtype=torch.int32, device=dev)
    dv = torch.empty((B, cap), dtype=torch.uint8, device=dev)
    if cap <= TOPK_MAX and USE_TOPK:
        _topk_gather(cs, cseg, cb, cl, img_hw, B, cap, DET_MIN_SIZE, db, ds, dl, dv)
    else:
        ks2, vs2 = _sort_desc(cs, cseg, B, 0)
        _lib.call('sgg_gather_topk', ks2.data_ptr(), vs2.data_ptr(), cseg.data_ptr(), cb.data_ptr(), cl.data_ptr(), img_hw.data_ptr(), B,
                  cap, DET_MIN_SIZE, db.data_ptr(), ds.data_ptr(), dl.data_ptr(), dv.data_ptr(), stream)
    mk = int(det.roi_heads.detections_per_img)
    kidx, kcnt = _nms(db, dl, dv, float(det.roi_heads.nms_thresh), mk)
    ob = torch.zeros((B, mk, 4), dtype=torch.float32, device=dev)
    osc = torch.zeros((B, mk), dtype=torch.float32, device=dev)
    ol = torch.zeros((B, mk), dtype=torch.int64, device=dev)
    _lib.call('sgg_det_output', db.data_ptr(), ds.data_ptr(), dl.data_ptr(), kidx.data_ptr(), kcnt.data_ptr(), B, cap, mk,
              ob.data_ptr(), osc.data_ptr(), ol.data_ptr(), stream)
    cnt = kcnt.cpu().tolist()
    out = []
    for b in range(B):
        n = cnt[b]
        bx = ob[b, :n]
        rh, rw = float(orig_sizes[b][0]) / float(image_sizes[b][0]), float(orig_sizes[b][1]) / float(image_sizes[b][1])
        scale = torch.tensor([rw, rh, rw, rh], dtype=torch.float32, device=dev)     # [3P] transform.postprocess
        out.append((bx, bx * scale, ol[b, :n], osc[b, :n]))
    return out
