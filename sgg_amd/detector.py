"""Frozen VGG-16 Faster-R-CNN front end (feature extractor part) on the HIP path.

Mirrors the module tree of the torchvision `FasterRCNN(vgg.features, ...)` object the reference builds at
sgg_models/rel_model_base.py:92-108, so that `state_dict()` keys are identical
(`detector.backbone.{0,2,5,...}.{weight,bias}`, `detector.rpn.head.*`, `detector.roi_heads.box_head.fc{6,7}.*`,
`detector.roi_heads.box_predictor.{cls_score,bbox_pred}.*`) and reference checkpoints load
(lib/pytorch_misc.py:183-203).  torchvision itself is not used (it is not installed); the nn.Modules below only
HOLD parameters -- the arithmetic runs in libsgg_hip.so.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops

VGG16_CFG = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512)


def make_vgg_features():
    """[3P] torchvision vgg16().features with module '30' (last max-pool) deleted (rel_model_base.py:312)."""
    layers, cin = [], 3
    for v in VGG16_CFG:
        if v == 'M':
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    seq = nn.Sequential(*layers)
    seq.out_channels = 512
    return seq


def make_vgg_classifier(in_dim=512 * 7 * 7, dim=4096):
    """[3P] vgg16().classifier with '6' deleted (rel_model_base.py:313): Linear ReLU Dropout Linear ReLU Dropout."""
    return nn.Sequential(nn.Linear(in_dim, dim), nn.ReLU(True), nn.Dropout(), nn.Linear(dim, dim), nn.ReLU(True),
                         nn.Dropout())


class _RPNHead(nn.Module):
    def __init__(self, c, a):
        super(_RPNHead, self).__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)
        self.cls_logits = nn.Conv2d(c, a, 1)
        self.bbox_pred = nn.Conv2d(c, a * 4, 1)


class _RPN(nn.Module):
    def __init__(self, c, a):
        super(_RPN, self).__init__()
        self.head = _RPNHead(c, a)


class _TwoMLPHead(nn.Module):
    def __init__(self, i, d):
        super(_TwoMLPHead, self).__init__()
        self.fc6 = nn.Linear(i, d)
        self.fc7 = nn.Linear(d, d)


class _Predictor(nn.Module):
    def __init__(self, d, ncls):
        super(_Predictor, self).__init__()
        self.cls_score = nn.Linear(d, ncls)
        self.bbox_pred = nn.Linear(d, ncls * 4)


class _RoIHeads(nn.Module):
    def __init__(self, c, pool, d, ncls, score_thresh, dets):
        super(_RoIHeads, self).__init__()
        self.box_head = _TwoMLPHead(c * pool * pool, d)
        self.box_predictor = _Predictor(d, ncls)
        self.score_thresh, self.nms_thresh, self.detections_per_img = score_thresh, 0.5, dets


class Transform(object):
    """[3P] GeneralizedRCNNTransform(min_size, max_size, ImageNet mean/std), size_divisible=32."""

    def __init__(self, min_size, max_size):
        self.min_size, self.max_size, self.size_divisible = min_size, max_size, 32

    def resized_hw(self, h, w):
        scale = min(float(self.min_size) / min(h, w), float(self.max_size) / max(h, w))
        return int(math.floor(h * scale)), int(math.floor(w * scale))


def is_u8_image(im):
    """decoded image as the dataset keeps it: uint8 [h, w, 3], torch tensor or numpy array"""
    return im.dtype in (torch.uint8, np.uint8) and im.ndim == 3 and im.shape[-1] == 3


def image_hw(im):
    """(h, w) the transform sees: the tensor's own size, or the SquarePad-ed size of a decoded u8 HWC image."""
    if is_u8_image(im):
        s = int(max(im.shape[0], im.shape[1]))
        return s, s
    return int(im.shape[-2]), int(im.shape[-1])


class VGGDetector(nn.Module):
    """Holds the detector parameters and runs transform + backbone (SURVEY a-1, a-2) through the C ABI."""

    def __init__(self, num_classes, min_size, max_size, pool_sz=7, obj_dim=4096, box_score_thresh=0.2,
                 box_detections_per_img=50):
        super(VGGDetector, self).__init__()
        self.backbone = make_vgg_features()
        self.rpn = _RPN(512, 15)  # 5 sizes x 3 ratios (rel_model_base.py:94-95)
        self.roi_heads = _RoIHeads(512, pool_sz, obj_dim, num_classes, box_score_thresh, box_detections_per_img)
        self.transform = Transform(min_size, max_size)
        self.mode = 'gtbox'
        self._prep = {}
        self._bufs = {}

    # ---- weights in kernel layout: conv1_1 [64,27] f32 (ky,kx,c); others [Cout,3,3,Cin] in the compute dtype
    def prepared(self, dtype):
        convs = [m for m in self.backbone if isinstance(m, nn.Conv2d)]
        key = (dtype,) + tuple((c.weight.data_ptr(), c.weight._version, c.bias._version) for c in convs)
        if self._prep.get('key') != key:
            ws = []
            for i, c in enumerate(convs):
                w = c.weight.detach().float()
                co, ci = w.shape[0], w.shape[1]
                wk = ops.permute_ncp_to_npc(w.reshape(co, ci, 9), torch.float32 if i == 0 else dtype)  # [co,9,ci]
                ws.append((wk.reshape(co, 9 * ci), c.bias.detach().float().contiguous(), ci, co))
            if ops.is_half(dtype) and ws[0][0].is_cuda:
                ws[0] = ws[0] + (ops.conv1_pack_weights(ws[0][0], dtype),)      # conv1_1 as MFMA fragments (the fused first block)
            self._prep = dict(key=key, val=ws)
        return self._prep['val']

    def _buf(self, name, shape, dtype, device, zero):
        k = (name, tuple(shape), dtype, str(device))
        b = self._bufs.get(k)
        if b is None:
            # zero-bordered planes are allocated (and zeroed) once; kernels only ever write interiors
            b = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=device)
            self._bufs[k] = b
        return b

    def _features_x3(self, x0, ws, B, Hp, Wp, dev, out):
        """The x3 mode's VGG-16 on PAIR planes (include/sgg_hip.h SGG_PAIR16: every activation as two f16 planes hi + lo, 22 significand
        bits): conv1_1 in fp32 arithmetic writes the first pair plane, every 3x3 convolution reads a pair plane and writes the next one
        from its epilogue (fp32 accumulator -> ReLU -> hi / lo), the pools run on pair planes, the last convolution writes the fp32 map
        RoIAlign reads.  No split pass over any activation, no [hi | hi | lo] copy (round 5: 14 sgg_split3 launches, ~1 ms per batch of 8)."""
        f16 = torch.float16
        cfg = list(VGG16_CFG)
        H, W = Hp, Wp
        w, bias = ws[0][:2]
        x = self._buf('p0', (B, H + 2, W + 2, 128), f16, dev, True)
        ops.conv1_1(x0, w, bias, x, pair=True)
        ci_layer, n_conv = 1, len(ws)
        skip_pool = False
        for li in range(1, len(cfg)):
            v = cfg[li]
            if v == 'M' and skip_pool:            # (rode in the previous convolution's epilogue)
                skip_pool = False
                continue
            if v == 'M':
                C = x.shape[3] // 2
                y = self._buf('p%d' % li, (B, H // 2 + 2, W // 2 + 2, 2 * C), f16, dev, True)
                ops.maxpool2x2_pair(x, y, 1)
                H, W = H // 2, W // 2
            else:
                w, bias, ci, co = ws[ci_layer][:4]
                last = ci_layer == n_conv - 1
                if last:
                    if out is not None:
                        assert tuple(out.shape) == (B, H, W, co) and out.dtype == torch.float32 and out.is_contiguous()
                    y = out if out is not None else torch.empty((B, H, W, co), dtype=torch.float32, device=dev)
                    ops.conv3x3_relu_pair(x, w.view(co, 3, 3, ci), bias, y, 0, pair_out=False)
                elif ops.conv_pp_x3_ok(H, W, ci, co):
                    # conv2_1 .. conv4_3: the LDS-resident patch kernel's x3 form, the block's MaxPool2d(2) in its epilogue
                    pool = li + 1 < len(cfg) and cfg[li + 1] == 'M' and H % 2 == 0 and W % 2 == 0
                    if pool:
                        y = self._buf('p%d' % (li + 1), (B, H // 2 + 2, W // 2 + 2, 2 * co), f16, dev, True)
                        H, W = H // 2, W // 2
                        skip_pool = True
                    else:
                        y = self._buf('p%d' % li, (B, H + 2, W + 2, 2 * co), f16, dev, True)
                    ops.conv3x3_relu_x3pp(x, w.view(co, 3, 3, ci), bias, y, 1, pool=pool)
                else:
                    y = self._buf('p%d' % li, (B, H + 2, W + 2, 2 * co), f16, dev, True)
                    ops.conv3x3_relu_pair(x, w.view(co, 3, 3, ci), bias, y, 1, pair_out=True)
                ci_layer += 1
            x = y
        return x

    def features(self, images, dtype, out=None):
        """images: list of f32[3,h,w] tensors in [0,1] (host or device) -- what the reference's SquarePad + ToTensor
        produce -- or of u8[h0,w0,3] decoded images, for which those two steps run inside the prep kernel.
        Returns (fmap NHWC [B,Hf,Wf,512] in `dtype`, image_sizes [(h,w)] after resize, (Hp,Wp) padded size).
        out (optional): the [B,Hf,Wf,512] tensor the last convolution writes (a caller that replays the head as a hipGraph keeps the map
        at one address, sgg_amd/graph_step.py).  `self._features_override` (same caller): a ready (fmap, sizes, padded) triple returned
        as it is -- the map was computed by another launch sequence."""
        ov = getattr(self, '_features_override', None)
        if ov is not None:
            return ov
        dev = self.backbone[0].weight.device
        sizes = [self.transform.resized_hw(*image_hw(im)) for im in images]
        d = self.transform.size_divisible
        Hp = int(math.ceil(max(s[0] for s in sizes) / d) * d)
        Wp = int(math.ceil(max(s[1] for s in sizes) / d) * d)
        B = len(images)
        ws = self.prepared(dtype)
        x0 = self._buf('img', (B, Hp + 2, Wp + 2, 4), torch.float32, dev, True)
        # the prep kernel writes image interiors only: the pad region of a previous, LARGER image in the same slot must be cleared --
        # i.e. only when the per-image sizes differ from the last call's on this (cached) plane
        sig = (tuple(sizes), x0.data_ptr())
        if torch.cuda.is_current_stream_capturing():
            # a launch sequence that will be replayed (sgg_amd/graph_step.py, graph_forward.py) cannot know what ran before it: it clears the plane
            # every time (48 MB at 8 frames: ~10 us), and the next plain call does too
            x0.zero_()
            self._x0_sig = ('captured',)
        elif getattr(self, '_x0_sig', None) != sig:
            if getattr(self, '_x0_sig', None) is not None:
                x0.zero_()
            self._x0_sig = sig
        staged = []
        for im in images:
            if is_u8_image(im):
                if isinstance(im, np.ndarray):
                    im = torch.from_numpy(np.ascontiguousarray(im))
                if not im.is_cuda:
                    im = im.to(device=dev, non_blocking=True)
            else:
                im = im.squeeze()
                if im.dtype != torch.float32 or not im.is_cuda:
                    im = im.to(device=dev, dtype=torch.float32, non_blocking=True)
            staged.append(im)
        ops.image_prep_batch(staged, sizes, x0)          # one launch for the batch (was one per image)
        if dtype == torch.float32 and ops.split3_on() and ops.PAIR_GEMM and getattr(self, '_features_split', None) is None:
            return self._features_x3(x0, ws, B, Hp, Wp, dev, out), sizes, (Hp, Wp)
        H, W = Hp, Wp
        x, ci_layer, n_conv = x0, 0, len(ws)
        cfg = list(VGG16_CFG)
        li = 0
        split = getattr(self, '_features_split', None)    # (sgg_amd/graph_step.py: told after every layer how many convolutions are done)
        told = 0
        while li < len(cfg):
            if split is not None and ci_layer != told:
                told = ci_layer
                split(ci_layer)
            v = cfg[li]
            if v == 'M':
                y = self._buf('a%d' % li, (B, H // 2 + 2, W // 2 + 2, x.shape[3]), dtype, dev, True)
                ops.maxpool2x2(x, y, 1)
                H, W = H // 2, W // 2
            else:
                w, bias, ci, co = ws[ci_layer][:4]
                last = ci_layer == n_conv - 1
                op = 0 if last else 1
                if (ci_layer == 0 and ops.is_half(dtype) and li + 2 < len(cfg) and cfg[li + 1] == 64 and cfg[li + 2] == 'M' and
                        ops.conv_pool_fusable(H, W, 64) and os.environ.get('SGG_CONV1_FUSE', '1') != '0'):
                    # the first block in one launch: conv1_1's full-resolution 64-channel map (the largest activation of the network) is
                    # computed tile by tile inside conv1_2's LDS patch and never written (csrc/conv_spatial.hip FUSE1)
                    w2, bias2 = ws[1][:2]
                    y = self._buf('a%d' % (li + 2), (B, H // 2 + 2, W // 2 + 2, 64), dtype, dev, True)
                    ops.conv1_block(x, ws[0][4], bias, w2.view(64, 3, 3, 64), bias2, y, 1, pool=True)
                    H, W = H // 2, W // 2
                    ci_layer += 2
                    li += 3
                    x = y
                    continue
                # conv followed by MaxPool2d(2): the pool rides in the conv epilogue (the full-resolution map is never written)
                fuse = (ci_layer > 0 and li + 1 < len(cfg) and cfg[li + 1] == 'M' and ops.conv_pool_fusable(H, W, co))
                if fuse:
                    y = self._buf('a%d' % (li + 1), (B, H // 2 + 2, W // 2 + 2, co), dtype, dev, True)
                    ops.conv3x3_relu(x, w.view(co, 3, 3, ci), bias, y, 1, pool=True)
                    H, W = H // 2, W // 2
                    li += 1
                else:
                    # the final map is handed to the caller (Result.fmap): a fresh tensor, never a cached plane
                    if last and out is not None:
                        assert tuple(out.shape) == (B, H, W, co) and out.dtype == dtype and out.is_contiguous(), (out.shape, (B, H, W, co))
                    y = (out if out is not None else torch.empty((B, H, W, co), dtype=dtype, device=dev)) if last else \
                        self._buf('a%d' % li, (B, H + 2, W + 2, co), dtype, dev, True)
                    if ci_layer == 0:
                        ops.conv1_1(x, w, bias, y)
                    else:
                        ops.conv3x3_relu(x, w.view(co, 3, 3, ci), bias, y, op)
                ci_layer += 1
            x = y
            li += 1
        return x, sizes, (Hp, Wp)
