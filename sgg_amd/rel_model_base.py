"""RelModelBase on the HIP path -- mirror of sgg_models/rel_model_base.py (same constructor, attributes, method
names, state_dict keys and exceptions), with the arithmetic in libsgg_hip.so.

Layout note: the feature map is channels-last in memory and handed out as a [B,C,H,W]-shaped view of NHWC storage
(the same logical tensor the reference returns).  RoI features (`node_feat`, `edge_feat`) are plain contiguous
[R,C,7,7] tensors exactly as in the reference (RoIAlign transposes through LDS), so fc6 weights are used un-permuted.
"""
import math
import os

import torch
import torch.nn as nn

from . import ops
from .detector import VGGDetector, image_hw, make_vgg_classifier
from .result import Result
from .union_boxes import UnionBoxesAndFeats

IM_SCALE = 592            # config.py:31
REL_FG_FRACTION = 0.25    # config.py:33


def enumerate_by_image_host(im_inds_host):
    """lib/pytorch_misc.py:493-502 on an already-host list of image indices: yields (img, start, end)."""
    s, cur = 0, int(im_inds_host[0])
    for i, val in enumerate(im_inds_host):
        if int(val) != cur:
            yield cur, s, i
            cur, s = int(val), i
    yield cur, s, len(im_inds_host)


def to_device_with_mirror(t_host, device, non_blocking=True):
    """Index tensors of the batch (gt_classes, gt_rels) are born on the host (dataloaders/blob.py builds them from numpy):
    the device copy keeps a reference to its host original, so the forward's bookkeeping (boxes per image, number of
    relation rows) reads the host copy instead of synchronising the stream for a D2H copy of data the host already has.
    The mirror is only attached here and by DeviceStager; a tensor without one goes through the D2H path."""
    t_dev = t_host.to(device, non_blocking=non_blocking)
    t_dev._sgg_host = t_host
    return t_dev


def host_of(t):
    """Host copy of a small index tensor: the attached mirror if there is one (same shape), else a synchronising D2H."""
    m = getattr(t, '_sgg_host', None)
    if m is not None and tuple(m.shape) == tuple(t.shape):
        return m
    return t.detach().to('cpu')


def rels_host_facts(rows, sizes):
    """What the host can say about a batch's ground-truth relation rows [(img, subj, obj, pred)] given the boxes per image {img: count}
    (lib/proposal_assignments_gtbox.py:28-45 counts them on the device): regular -- every row names two different boxes of an image of
    the batch (otherwise the device count decides); fg_pairs -- distinct ordered (img, subj, obj); max_edges -- most label rows one image
    ends up with (its candidates - its distinct FG pairs + its FG relations); max_per_pair -- most rows on one unordered box pair."""
    fg, regular = set(), True
    for im, s_, o_, _ in rows:
        if im in sizes and s_ != o_ and 0 <= s_ < sizes[im] and 0 <= o_ < sizes[im]:
            fg.add((im, s_, o_))
        else:
            regular = False
    max_edges = max_per_pair = None
    if regular:
        per_im = {i: sz * (sz - 1) for i, sz in sizes.items()}
        for im, _, _ in fg:
            per_im[im] -= 1
        fg_rows = {}
        for im, s_, o_, _ in rows:
            per_im[im] += 1
            fg_rows[(im, s_, o_)] = fg_rows.get((im, s_, o_), 0) + 1
        max_edges = max(per_im.values())
        max_per_pair = max([2] + [c + max(1, fg_rows.get((im, o_, s_), 0)) for (im, s_, o_), c in fg_rows.items()])
    return dict(regular=regular, fg_pairs=len(fg), max_edges=max_edges, max_per_pair=max_per_pair)


def as_nchw_view(x_nhwc):
    return x_nhwc.permute(0, 3, 1, 2)


def to_nhwc(x, dtype):
    """[B,C,H,W]-shaped feature map (a view of NHWC storage, or a plain NCHW tensor from another producer, e.g. the
    GAN of main.py:141-149) -> contiguous [B,H,W,C] in `dtype`."""
    R, C, Ph, Pw = x.shape
    v = x.permute(0, 2, 3, 1)
    if v.is_contiguous():
        return v if v.dtype == dtype else ops.cast(v, dtype)
    if x.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        x = x.float()
    return ops.permute_ncp_to_npc(x.reshape(R, C, Ph * Pw), dtype).view(R, Ph, Pw, C)


def to_rows(x, dtype):
    """RoI features [R,C,P,P] (any strides / float dtype) -> contiguous [R, C*P*P] in `dtype` (fc6's A operand)."""
    R = x.shape[0]
    if x.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        x = x.float()
    x = x.contiguous()                      # no-op for the tensors RoIAlign produced
    if x.dtype != dtype:
        x = x.to(dtype) if (x.requires_grad and torch.is_grad_enabled()) else ops.cast(x, dtype)   # keep a caller's graph intact
    return x.view(R, -1)


class _RoIFeatures(torch.autograd.Function):
    """node + union-box RoIAlign as one differentiable op in `fmap` (callers that keep the feature map in the graph: the GAN
    feature-augmentation path, main.py:141-149).  The hot path's fmap is detached (rel_model_stanford.py:131) and never gets here."""

    @staticmethod
    def forward(ctx, fmap, rois, union_inds, scale, P, dtype):
        fm = to_nhwc(fmap.detach(), dtype)
        node = ops.roi_align(fm, rois, None, scale, P, 2)
        edge = ops.roi_align(fm, rois, union_inds, scale, P, 2)
        ctx.save_for_backward(rois, union_inds)
        ctx.meta = (tuple(fm.shape), scale, fmap.dtype)
        return node, edge

    @staticmethod
    def backward(ctx, d_node, d_edge):
        rois, union_inds = ctx.saved_tensors
        shape, scale, in_dtype = ctx.meta
        prep = lambda g: g.contiguous() if g.dtype in (torch.float32, torch.bfloat16, torch.float16) else g.float().contiguous()
        d_fm = ops.roi_align_bwd(prep(d_node), shape, rois, None, scale)
        ops.roi_align_bwd(prep(d_edge), shape, rois, union_inds, scale, d_fmap=d_fm)
        return d_fm.permute(0, 3, 1, 2).to(in_dtype), None, None, None, None, None


_PARITY_SAID = {}


def _say_parity(clause):
    """once per process and mode: the default is the fast 16-bit mode, and a caller following INTEGRATION.md should know
    what it costs before comparing logits with the reference's"""
    if clause['logits_within_1e-3'] or _PARITY_SAID.get(clause['mode']):
        return
    _PARITY_SAID[clause['mode']] = True
    import logging
    logging.getLogger('sgg_amd').warning(
        'sgg_amd: compute mode %s -- %s.  For logits within 1e-3 of the fp32 reference: model.set_compute_dtype(torch.float32, split3=True) '
        '(x3 mode) or model.set_compute_dtype(torch.float32) (exact).', clause['mode'], clause['note'])


class RelModelBase(nn.Module):
    """RELATIONSHIPS (sgg_models/rel_model_base.py:22-123)."""

    def __init__(self, train_data, mode='sgcls', require_overlap_det=True, use_bias=False, test_bias=False,
                 backbone='vgg16', RELS_PER_IMG=1024, min_size=None, max_size=None, edge_model='motifs'):
        super(RelModelBase, self).__init__()
        self.classes = train_data.ind_to_classes
        self.rel_classes = train_data.ind_to_predicates
        self.mode = mode
        self.backbone = backbone
        self.RELS_PER_IMG = RELS_PER_IMG
        self.pool_sz = 7
        self.stride = 16
        self.use_bias = use_bias
        self.test_bias = test_bias
        self.require_overlap = require_overlap_det and self.mode == 'sgdet'
        if self.backbone == 'vgg16':
            self.obj_dim = 4096
            self.fmap_sz = 38
            min_size = IM_SCALE if min_size is None else min_size
            max_size = IM_SCALE if max_size is None else max_size
            self.detector = VGGDetector(len(self.classes), min_size, max_size, self.pool_sz, self.obj_dim)
            # rel_model_base.py:110-111: roi_fmap = Flatten + [fc6,ReLU,Dropout,fc7]; roi_fmap_obj = full classifier
            cls = make_vgg_classifier(512 * self.pool_sz ** 2, self.obj_dim)
            del cls._modules['5']
            del cls._modules['4']
            self.roi_fmap = nn.Sequential(nn.Flatten(), cls)
            self.roi_fmap_obj = make_vgg_classifier(512 * self.pool_sz ** 2, self.obj_dim)
        elif self.backbone == 'resnet50':
            # GQA / Mask-R-CNN-R50-FPN detector (rel_model_base.py:58-81): sgg_amd/resnet_fpn.py
            from .resnet_fpn import ResNet50FPNDetector, make_box_head
            self.obj_dim = 1024
            self.fmap_sz = 21
            min_size = 1333 if min_size is None else min_size
            max_size = 1333 if max_size is None else max_size
            self.detector = ResNet50FPNDetector(len(self.classes), min_size, max_size, self.pool_sz, self.obj_dim)
            self.roi_fmap = make_box_head(self.pool_sz, self.obj_dim)            # :78-79 deep copies of the detector's TwoMLPHead
            self.roi_fmap_obj = make_box_head(self.pool_sz, self.obj_dim)
        else:
            raise NotImplementedError(self.backbone)
        self.edge_dim = self.detector.backbone.out_channels
        self.union_boxes = UnionBoxesAndFeats(pooling_size=self.pool_sz, stride=self.stride, dim=self.edge_dim,
                                              edge_model=edge_model)
        if self.use_bias:
            from .sparse_targets import FrequencyBias
            self.freq_bias = FrequencyBias(train_data)                                # rel_model_base.py:120-121
        # HIP-path settings (not in the reference): storage/compute type of activations and weights.
        self.compute_dtype = torch.float16       # see set_compute_dtype
        _say_parity(self.parity_clause)
        self._prep = {}
        self._shadow, self._shadow_tags = {}, {}     # compute-dtype weight copies (rel_model_stanford._shadow_cast)

    def fc_layers(self):
        """The four big Linear layers of the RoI feature heads under the names their parameters have in the reference's state_dict
        -- {'fc6_edge' | 'fc7_edge' | 'fc6_obj' | 'fc7_obj': (parameter-name prefix, module)} -- and whether the edge branch's fc7 is
        followed by a ReLU: vgg16 keeps the VGG classifier (roi_fmap = Flatten + [fc6, ReLU, Dropout, fc7]: no ReLU after fc7,
        rel_model_base.py:110-111), resnet50 copies the detector's TwoMLPHead (relu(fc7), :78-80)."""
        if self.backbone == 'resnet50':
            return {'fc6_edge': ('roi_fmap.fc6', self.roi_fmap.fc6), 'fc7_edge': ('roi_fmap.fc7', self.roi_fmap.fc7),
                    'fc6_obj': ('roi_fmap_obj.fc6', self.roi_fmap_obj.fc6), 'fc7_obj': ('roi_fmap_obj.fc7', self.roi_fmap_obj.fc7)}, True
        return {'fc6_edge': ('roi_fmap.1.0', self.roi_fmap[1][0]), 'fc7_edge': ('roi_fmap.1.3', self.roi_fmap[1][3]),
                'fc6_obj': ('roi_fmap_obj.0', self.roi_fmap_obj[0]), 'fc7_obj': ('roi_fmap_obj.3', self.roi_fmap_obj[3])}, False

    # ------------------------------------------------------------------ reference API
    @property
    def num_classes(self):
        return len(self.classes)

    @property
    def num_rels(self):
        return len(self.rel_classes)

    def predict(self, node_feat, edge_feat, rel_inds, rois, im_sizes):
        raise NotImplementedError('predict')

    def forward(self, batch):
        raise NotImplementedError('forward')

    def set_compute_dtype(self, dtype, split3=False, backward_f16=False):
        """torch.float32 = exact-fp32 MFMA mode (the reference's own precision: the 1e-3 parity bar); torch.float16 / torch.bfloat16 =
        16-bit storage and MFMA operands with fp32 accumulation -- the same kernels at the same rates.  float16 (11-bit significand) is the
        throughput mode whose logits stay within 0.1 / 0.03 of the reference's (DESIGN.md "f16"); bfloat16 (8 bits, wider exponent) is kept
        as BASELINE.json words its configuration: 8 times the rounding error, no loss scaling needed in training."""
        ops.dt(dtype)
        if split3 and dtype != torch.float32:
            raise ValueError('split3 (the x3 mode) is a form of the fp32 mode: set_compute_dtype(torch.float32, split3=True)')
        if dtype != self.compute_dtype or bool(split3) != getattr(self, 'split3', False):
            self._operand_generation = getattr(self, '_operand_generation', 0) + 1     # derived operands get new buffers: captured launch sequences (graph_step.py) are void
        self.compute_dtype = dtype
        # x3 mode (sgg_amd/ops.py "the x3 mode"): fp32 storage and element-wise arithmetic as in the exact-fp32 mode, every MFMA contraction on
        # f16 split operands (hi + lo, three products, fp32 accumulate): fp32-grade logits (within the 1e-3 clause) at several times the
        # exact mode's rate.  The switch is process-wide (ops.set_split3): one compute mode at a time.
        self.split3 = bool(split3)
        # x3 with backward_f16=True: the training BACKWARD's contractions on operands rounded to f16 once (one MFMA product instead of three,
        # under the Trainer's loss scale): the forward -- what both parity clauses are about -- stays fp32-grade, the gradients have the f16
        # mode's accuracy (ops.set_backward_f16).  Mixed-precision training with an fp32-grade forward.
        if backward_f16 and not split3:
            raise ValueError('backward_f16 is an option of the x3 mode: set_compute_dtype(torch.float32, split3=True, backward_f16=True)')
        self.backward_f16 = bool(backward_f16)
        ops.set_split3(self.split3)
        _say_parity(self.parity_clause)
        return self

    @property
    def parity_clause(self):
        """Which of the north star's two parity clauses the CURRENT compute mode meets on the benchmark configuration (8 x 592x592, 32 boxes;
        measured by tests/test_parity_full_gpu.py, records under profiles/r0x_parity_bench_config.json / r0x_recall_parity.json):
        rel_dists / obj_dists within 1e-3 of the fp32 reference, and R@K within +-0.1 points."""
        dt = self.compute_dtype
        if dt == torch.float32:
            mode = 'x3' if getattr(self, 'split3', False) else 'f32'
            return {'mode': mode, 'logits_within_1e-3': True, 'recall_within_0.1': True,
                    'note': 'fp32-grade: logits within 4e-4 (x3) / 8e-5 (exact fp32) of the reference'}
        if dt == torch.float16:
            return {'mode': 'f16', 'logits_within_1e-3': False, 'recall_within_0.1': True,
                    'note': 'f16 operands: logits within ~0.05 (obj) / 0.01 (rel) of the fp32 reference -- the 1e-3 clause is NOT met; R@K within 0.05 points (met)'}
        return {'mode': 'bf16', 'logits_within_1e-3': False, 'recall_within_0.1': False,
                'note': 'bf16 operands: logits within ~0.44 (obj) / 0.10 (rel) of the fp32 reference and R@K up to 0.2 points off -- NEITHER parity clause is met'}

    def spatial_scale(self, im_sizes):
        """[3P] MultiScaleRoIAlign.infer_scale: 2^round(log2(fmap_size / image_size)); 1/16 at the configs."""
        size = max(max(s[0], s[1]) for s in im_sizes)
        return 2.0 ** round(math.log2(float(self.fmap_hw[0]) / float(size))) if hasattr(self, 'fmap_hw') else 1.0 / 16

    def get_rel_inds(self, rel_labels, im_inds, box_priors, _num=None):
        """rel_model_base.py:143-165.  `_num` (private): the pair count when the caller already knows it on the host
        (all same-image ordered pairs, no overlap filter), which saves the device->host read of the count."""
        if self.training:
            return rel_labels[:, :3].data.clone()
        out, count = ops.pair_index_eval(im_inds.contiguous(), box_priors.float().contiguous() if self.require_overlap
                                         else None, self.require_overlap)
        n = _num if (_num is not None and not self.require_overlap) else int(count.item())
        if n == 0:
            return im_inds.new_zeros((1, 3))  # :160-161
        return out[:n]

    def set_box_score_thresh(self, box_score_thresh):
        self.detector.roi_heads.score_thresh = box_score_thresh  # rel_model_base.py:168-172

    def faster_rcnn(self, x, gt_boxes, gt_classes, gt_rels):
        """rel_model_base.py:175-242: gt-box branch (predcls / sgcls) and the detector branch (sgdet)."""
        if self.mode == 'sgdet':
            return self._faster_rcnn_sgdet(x, gt_classes)
        dtype = self.compute_dtype
        im_host = host_of(gt_classes)[:, 0].tolist()           # a D2H sync only when the batch carries no host mirror
        segs = list(enumerate_by_image_host(im_host))
        images = [x[i] for i, _, _ in segs]                              # :180 (x is indexed by image id)
        fmap, sizes, _ = self.detector.features(images, dtype)           # :183-184
        self.fmap_hw = (fmap.shape[1], fmap.shape[2])
        # boxes scaled by the resize ratio, per image ([3P] resize_boxes)
        ratios = []
        for (i, s, e), (nh, nw) in zip(segs, sizes):
            h, w = image_hw(x[i])
            ratios.append((s, e, float(nw) / float(w), float(nh) / float(h)))
        gt_boxes = gt_boxes.float()
        if all(r[2] == 1.0 and r[3] == 1.0 for r in ratios):
            priors = gt_boxes.clone()                            # rel_model_base.py:197: a copy -- a consumer that rescales result.rm_box_priors in place must not edit the batch's gt_boxes
        else:
            scale = torch.ones((gt_boxes.shape[0], 4), dtype=torch.float32)
            for s, e, rw, rh in ratios:
                scale[s:e] = torch.tensor([rw, rh, rw, rh])
            priors = gt_boxes * scale.to(gt_boxes.device, non_blocking=True)
        # what depends on the boxes-per-image counts only lives on the device once per count signature: the image index per box
        # (i64, contiguous, and as an f32 column for the RoI rows) and the first box of every image -- a step launches nothing for them
        seg_t = self._segment_tensors(segs, im_host, gt_boxes.device)
        _, obj_labels, rel_labels = self.gt_labels(gt_boxes, gt_classes, gt_rels, segs=segs, _seg_t=seg_t)   # :189
        im_inds = seg_t['im_inds']
        result = Result(od_obj_labels=obj_labels, rm_box_priors=priors, rm_obj_labels=obj_labels,
                        rel_labels=rel_labels, im_inds=im_inds)
        result.rm_box_priors_org = gt_boxes
        result.im_sizes_org = [image_hw(x[i]) for i, _, _ in segs]
        result.im_sizes = sizes
        result.fmap = as_nchw_view(fmap)
        result.rois = torch.cat((seg_t['im_col'], priors), 1)
        result._num_pairs = sum((e - s) * (e - s - 1) for _, s, e in segs)   # host-side count (private)
        result._segs = segs                                                    # (image, first box, end box): host-side (private)
        # host-side facts about the graphs (private): (images, most nodes, most candidate edges in one image).  In training
        # gt_labels knows the exact rows per image (duplicate FG relations add rows) when the batch has a host mirror.
        worst = getattr(rel_labels, '_sgg_max_edges', None) if rel_labels is not None else \
            max((e - s) * (e - s - 1) for _, s, e in segs)
        result._graphs = (segs[-1][0] + 1, max(e - s for _, s, e in segs), worst) if worst is not None else None
        return result

    def _faster_rcnn_sgdet(self, x, gt_classes):
        """rel_model_base.py:209-235: RPN + RoI heads (sgg_amd/sgdet.py), <= 50 detections per image."""
        from . import sgdet
        # The detector is frozen (main.py:62-63) and always runs as an inference detector here, also under model.train():
        # torchvision's RoIHeads in training mode returns losses and no detections, which is why the reference documents
        # SGDet training as unsupported (README.md:214-218).  rm_obj_labels are the detector's labels, as at :221,:228.
        dtype = self.compute_dtype
        if gt_classes is not None and gt_classes.numel() > 0:
            ids = sorted(set(int(v) for v in gt_classes[:, 0].detach().to('cpu').tolist()))
        else:
            ids = list(range(len(x)))
        images = [x[i] for i in ids]
        pyramid = None
        if self.backbone == 'resnet50':          # RPN over P2 .. P5 + pool, box head over P2 .. P5 ([3P] maskrcnn_resnet50_fpn)
            fmap, sizes, padded, pyramid = self.detector.features(images, dtype, pyramid=True)
        else:
            fmap, sizes, padded = self.detector.features(images, dtype)
        self.fmap_hw = (fmap.shape[1], fmap.shape[2])
        orig = [image_hw(im) for im in images]
        dets = sgdet.detect(self, fmap, sizes, padded, orig, self.spatial_scale(sizes), pyramid=pyramid)
        priors, priors_org, labels, im_inds = [], [], [], []
        for i, (bx, bx_org, lab, _) in enumerate(dets):
            if bx.shape[0] <= 1:
                raise ValueError('at least two objects must be detected to build relationships, make sure the detector is '
                                 'properly pretrained', dets)                                       # :216-219
            priors.append(bx)
            priors_org.append(bx_org)
            labels.append(lab)
            im_inds.append(torch.full((bx.shape[0],), i, dtype=torch.int64, device=bx.device))
        im_inds = torch.cat(im_inds)
        result = Result(rm_obj_labels=torch.cat(labels).view(-1), rm_box_priors=torch.cat(priors), rel_labels=None,
                        im_inds=im_inds)
        result.rm_box_priors_org = torch.cat(priors_org)
        if result.rm_box_priors.shape[0] <= 1:
            raise ValueError('at least two objects must be detected to build relationships')        # :234-235
        result.im_sizes_org = orig
        result.im_sizes = sizes
        result.fmap = as_nchw_view(fmap)
        result.rois = torch.cat((im_inds.float()[:, None], result.rm_box_priors), 1)
        return result

    def node_edge_features(self, fmap, rois, union_inds, im_sizes, _pairing=None):
        """rel_model_base.py:245-260: RoIAlign of the boxes and of the pair union boxes (union fused in-kernel).
        _pairing (sgg_amd/pairing.py, only forward() passes it): pool every unordered pair once; the edge features come back as a
        PairedEdgeFeat ([U,C,P,P] rows + the edge -> row map; `.dense()` = the [E,C,P,P] tensor)."""
        assert union_inds.shape[1] == 2, union_inds.shape
        dtype = self.compute_dtype
        fm = to_nhwc(fmap, dtype)
        self.fmap_hw = (fm.shape[1], fm.shape[2])
        scale = self.spatial_scale(im_sizes) if im_sizes is not None else 1.0 / self.stride
        rois = rois.float().contiguous()
        if fmap.requires_grad and torch.is_grad_enabled():
            return _RoIFeatures.apply(fmap, rois, union_inds.contiguous(), scale, self.pool_sz, dtype)
        node = ops.roi_align(fm, rois, None, scale, self.pool_sz, 2)
        if _pairing is not None:
            from .pairing import PairedEdgeFeat
            return node, PairedEdgeFeat(ops.roi_align(fm, rois, _pairing.pairs, scale, self.pool_sz, 2), _pairing)
        edge = ops.roi_align(fm, rois, union_inds.contiguous(), scale, self.pool_sz, 2)
        return node, edge

    def get_scaled_boxes(self, boxes, im_inds, im_sizes):
        """rel_model_base.py:263-274."""
        boxes_scaled = boxes.clone()
        for im_ind, s, e in enumerate_by_image_host(im_inds.long().to('cpu').tolist()):
            boxes_scaled[s:e, [0, 2]] = boxes_scaled[s:e, [0, 2]] / im_sizes[im_ind][1]
            boxes_scaled[s:e, [1, 3]] = boxes_scaled[s:e, [1, 3]] / im_sizes[im_ind][0]
        assert boxes_scaled.max() <= 1 + 1e-3, (boxes_scaled.max(), boxes.max(), im_sizes)
        return boxes_scaled

    def _segment_tensors(self, segs, im_host, dev):
        key = (tuple((i, e - s) for i, s, e in segs), str(dev))
        cache = self.__dict__.setdefault('_seg_cache', {})
        t = cache.get(key)
        if t is None:
            if len(cache) > 64:
                cache.clear()
            im = torch.tensor(im_host, dtype=torch.int64).to(dev)
            first = [0] * (segs[-1][0] + 1)
            for i, s, _ in segs:
                first[i] = s
            t = cache[key] = dict(im_inds=im, im_col=im.float()[:, None].contiguous(),
                                  first=torch.tensor(first, dtype=torch.int32).to(dev))
        return t

    def gt_labels(self, gt_boxes, gt_classes, gt_rels=None, sample_factor=-1, segs=None, _seg_t=None):
        """rel_model_base.py:277-300 + proposal_assignments_gtbox (lib/proposal_assignments_gtbox.py:7-80).
        `_seg_t` (private, faster_rcnn): the cached per-signature index tensors; the RoI rows are then left to the caller (None)."""
        assert gt_boxes is not None
        im_inds = gt_classes[:, 0]
        rois = torch.cat((im_inds.float()[:, None], gt_boxes.float()), 1) if _seg_t is None else None
        if gt_rels is not None and self.training:
            if segs is None:
                segs = list(enumerate_by_image_host(im_inds.to('cpu').tolist()))
            num_im = segs[-1][0] + 1
            first = [0] * num_im
            for i, s, _ in segs:
                first[i] = s
            n_cand = sum((e - s) * (e - s - 1) for _, s, e in segs)
            R = gt_rels.shape[0]
            cap = n_cand + R
            out, count = ops.pair_index_train(_seg_t['im_inds'] if _seg_t is not None else im_inds.long().contiguous(), gt_rels.long().contiguous(),
                                              _seg_t['first'] if _seg_t is not None else torch.tensor(first, dtype=torch.int32).to(im_inds.device), cap)
            # rows = candidates + (extra rows for duplicate FG relations on one pair); FG pairs replace a candidate.
            # With a host mirror of gt_rels the count is computed here (R is a few dozen); otherwise read it back.
            rels_host = getattr(gt_rels, '_sgg_host', None)
            if rels_host is not None and tuple(rels_host.shape) == tuple(gt_rels.shape):
                facts = rels_host_facts(rels_host.tolist(), {i: e - s for i, s, e in segs})
                regular, fg_pairs = facts['regular'], facts['fg_pairs']
                n = n_cand - fg_pairs + R if regular else int(count.item())
                max_edges, max_per_pair = facts['max_edges'], facts['max_per_pair']
            else:
                n = int(count.item())
                max_edges = None
            if __debug__ and os.environ.get('SGG_CHECK_COUNTS'):
                assert n == int(count.item()), ('host-side row count differs from the device count', n, int(count.item()))
            rel_labels = out[:n]
            # sub-sampling (lib/proposal_assignments_gtbox.py:47-66): at most RELS_PER_IMG*0.25*num_im FG rows and
            # RELS_PER_IMG*num_im rows in total (or num_fg*sample_factor BG rows).  random_choose is a uniform random
            # subset; rows keep their (img, subj, obj) order, so the final sort of :74-77 is already satisfied.
            num_fg = min(R, int(self.RELS_PER_IMG * REL_FG_FRACTION * num_im))
            n_bg_all = n - R
            sample_bg = num_im > 1 and sample_factor > -1
            num_bg = min(n_bg_all, int(num_fg * sample_factor) if sample_bg else int(self.RELS_PER_IMG * num_im) - num_fg)
            if num_fg < R or num_bg < n_bg_all:
                # FG rows are recognised by their label: the reference samples fg_rels by identity (:28-35), which is the same set
                # unless a GT relation carries predicate 0 (not a VG / GQA predicate id: those start at 1) -- such a row would
                # count as background here.  The row counts are taken from the masks themselves, never from R / the host count,
                # so the permutation always matches the rows it indexes.
                is_fg = rel_labels[:, 3] > 0
                keep = torch.ones(n, dtype=torch.bool, device=rel_labels.device)
                for mask, want in ((is_fg, num_fg), (~is_fg, max(num_bg, 0))):
                    pos = torch.nonzero(mask).view(-1)
                    have = int(pos.numel())
                    if want < have:
                        drop = pos[torch.randperm(have, device=pos.device)[want:]]
                        keep[drop] = False
                rel_labels = rel_labels[keep].contiguous()
            if max_edges is not None:
                rel_labels._sgg_max_edges = max_edges       # sub-sampling only removes rows: still an upper bound
                rel_labels._sgg_max_per_pair = max_per_pair
                # every ordered pair exactly once, nothing sub-sampled: the rows are the evaluation candidate list (plus labels), so
                # the graph index tables cached per box-count signature apply (rel_model_stanford.forward)
                rel_labels._sgg_plain = bool(R == fg_pairs and rel_labels.shape[0] == n_cand)
            obj_labels = gt_classes[:, 1]               # (a strided view: the fused cross-entropy and F.cross_entropy both take it)
        else:
            obj_labels = gt_classes[:, 1]
            rel_labels = None
        return rois, obj_labels, rel_labels
