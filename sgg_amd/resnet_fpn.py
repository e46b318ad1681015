"""Frozen ResNet-50-FPN front end of the GQA configuration (sgg_models/rel_model_base.py:58-81, config.py:76-78: `-data gqa` forces
`backbone='resnet50'`), feature-extractor part, on the HIP path.

The module tree mirrors the torchvision `maskrcnn_resnet50_fpn` object the reference builds (box predictor replaced, mask predictor
removed), so `state_dict()` keys are the reference's: `detector.backbone.body.{conv1,bn1,layerN.M.{conv1,bn1,conv2,bn2,conv3,bn3,
downsample.{0,1}}}`, `detector.backbone.fpn.{inner_blocks,layer_blocks}.K`, `detector.rpn.head.*`,
`detector.roi_heads.{box_head.fc6,box_head.fc7,box_predictor.*,mask_head.*}`.  torchvision is not used; the modules only HOLD
parameters.  PARITY UNPINNED [3P]: the arithmetic is torchvision's (ResNet v1.5 bottlenecks with the stride on the 3x3 convolution,
FrozenBatchNorm2d with eps 1e-5, FPN top level, LastLevelMaxPool), restated from its published definition.

What the relation model reads is `Result.fmap = fmap_multiscale[last key]` (rel_model_base.py:239): the 'pool' level -- P5 sub-sampled
by 2, 256 channels at stride 64 (fmap_sz 21 for 1333-pixel images) -- and P5 depends on C5 only (the top-down pathway starts there).
So the gt-box modes (predcls / sgcls) need: stem, layer1-4, the lateral 1x1 and output 3x3 convolution of level 3, the sub-sampling.
Every convolution is a contraction on this package's MFMA kernels: 1x1 convolutions are GEMMs on the NHWC rows; the 7x7 stem and the
three stride-2 3x3 convolutions go through a patch matrix (`sgg_im2col`) + GEMM; the thirteen stride-1 3x3 convolutions run on the
spatial-conv kernel over zero-bordered planes.  FrozenBatchNorm is folded into the convolution weights when they are prepared.
sgdet (the detector branch, rel_model_base.py:209-235) needs the whole pyramid: `features(..., pyramid=True)` also returns P2 .. P5 --
lateral 1x1 convolutions of C2 .. C5, the top-down nearest-neighbour joins (`sgg_upsample_add`) and the four 3x3 output convolutions
-- for the five-level RPN and the four-level RoIAlign of sgg_amd/sgdet.py.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .detector import Transform, _Predictor, _RPN, _TwoMLPHead, image_hw, is_u8_image

LAYERS = ((64, 256, 3, 1), (128, 512, 4, 2), (256, 1024, 6, 2), (512, 2048, 3, 2))     # (mid, out, blocks, stride of the first block)
BN_EPS = 1e-5


class FrozenBatchNorm2d(nn.Module):
    """[3P] torchvision.ops.misc.FrozenBatchNorm2d: four buffers, no statistics update, y = (x - mean) * weight / sqrt(var + eps) + bias"""

    def __init__(self, n):
        super(FrozenBatchNorm2d, self).__init__()
        self.register_buffer('weight', torch.ones(n))
        self.register_buffer('bias', torch.zeros(n))
        self.register_buffer('running_mean', torch.zeros(n))
        self.register_buffer('running_var', torch.ones(n))

    def affine(self):
        scale = self.weight.float() * torch.rsqrt(self.running_var.float() + BN_EPS)
        return scale, self.bias.float() - self.running_mean.float() * scale


class _Bottleneck(nn.Module):
    def __init__(self, cin, mid, cout, stride):
        super(_Bottleneck, self).__init__()
        self.conv1, self.bn1 = nn.Conv2d(cin, mid, 1, bias=False), FrozenBatchNorm2d(mid)
        self.conv2, self.bn2 = nn.Conv2d(mid, mid, 3, stride=stride, padding=1, bias=False), FrozenBatchNorm2d(mid)
        self.conv3, self.bn3 = nn.Conv2d(mid, cout, 1, bias=False), FrozenBatchNorm2d(cout)
        self.stride = stride
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride=stride, bias=False), FrozenBatchNorm2d(cout))


class _Body(nn.Module):
    def __init__(self):
        super(_Body, self).__init__()
        self.conv1, self.bn1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False), FrozenBatchNorm2d(64)
        cin = 64
        for li, (mid, cout, blocks, stride) in enumerate(LAYERS):
            seq = nn.Sequential(*[_Bottleneck(cin if b == 0 else cout, mid, cout, stride if b == 0 else 1) for b in range(blocks)])
            setattr(self, 'layer%d' % (li + 1), seq)
            cin = cout


class _FPN(nn.Module):
    def __init__(self, out_channels=256):
        super(_FPN, self).__init__()
        self.inner_blocks = nn.ModuleList([nn.Conv2d(c, out_channels, 1) for _, c, _, _ in LAYERS])
        self.layer_blocks = nn.ModuleList([nn.Conv2d(out_channels, out_channels, 3, padding=1) for _ in LAYERS])


class _Backbone(nn.Module):
    def __init__(self):
        super(_Backbone, self).__init__()
        self.body = _Body()
        self.fpn = _FPN(256)
        self.out_channels = 256


class _MaskHead(nn.Module):
    """parameters only: the reference sets mask_predictor = None but keeps mask_head in the module tree (and so in checkpoints)"""

    def __init__(self, c=256):
        super(_MaskHead, self).__init__()
        for i in range(1, 5):
            setattr(self, 'mask_fcn%d' % i, nn.Conv2d(c, c, 3, padding=1))


class _RoIHeadsFPN(nn.Module):
    def __init__(self, pool, d, ncls, score_thresh, dets):
        super(_RoIHeadsFPN, self).__init__()
        self.box_head = _TwoMLPHead(256 * pool * pool, d)
        self.box_predictor = _Predictor(d, ncls)
        self.mask_head = _MaskHead(256)
        self.score_thresh, self.nms_thresh, self.detections_per_img = score_thresh, 0.5, dets


def make_box_head(pool_sz=7, dim=1024):
    """roi_fmap / roi_fmap_obj of the resnet50 configuration: deep copies of the detector's TwoMLPHead (rel_model_base.py:78-80):
    flatten -> relu(fc6) -> relu(fc7)."""
    return _TwoMLPHead(256 * pool_sz * pool_sz, dim)


class ResNet50FPNDetector(nn.Module):
    """Holds the detector parameters and runs transform + ResNet-50 + the top FPN level + LastLevelMaxPool through the C ABI."""

    def __init__(self, num_classes, min_size=1333, max_size=1333, pool_sz=7, obj_dim=1024, box_score_thresh=0.2,
                 box_detections_per_img=50):
        super(ResNet50FPNDetector, self).__init__()
        self.backbone = _Backbone()
        self.rpn = _RPN(256, 3)                      # one size x 3 ratios per level
        self.roi_heads = _RoIHeadsFPN(pool_sz, obj_dim, num_classes, box_score_thresh, box_detections_per_img)
        self.transform = Transform(min_size, max_size)
        self.mode = 'gtbox'
        self._prep = {}

    # ---- weights in kernel layout, FrozenBatchNorm folded in: 1x1 [Cout, Cin]; 3x3 stride 1 [Cout,3,3,Cin]; patch-matrix
    # convolutions [Cout, Kp] with columns (ky, kx, c) zero-filled to Kp
    def prepared(self, dtype):
        body, fpn = self.backbone.body, self.backbone.fpn
        key = (dtype, body.conv1.weight.data_ptr(), body.conv1.weight._version, fpn.layer_blocks[3].weight._version,
               body.layer4[2].conv3.weight._version, body.bn1.running_var._version)
        if self._prep.get('key') == key:
            return self._prep['val']

        def fold(conv, bn):
            w = conv.weight.detach().float()
            scale, shift = bn.affine()
            return w * scale.view(-1, 1, 1, 1), shift.contiguous()

        def rows(w, Kp=None):                       # [Cout, Cin, k, k] -> [Cout, k*k*Cin (-> Kp)] in (ky, kx, c) order
            co = w.shape[0]
            m = w.permute(0, 2, 3, 1).reshape(co, -1)
            if Kp is not None and Kp != m.shape[1]:
                m = torch.cat((m, m.new_zeros(co, Kp - m.shape[1])), 1)
            return m.contiguous().to(dtype)

        p = {}
        w, b = fold(body.conv1, body.bn1)
        p['stem'] = (rows(w, 192), b)
        blocks = []
        for li in range(4):
            for blk in getattr(body, 'layer%d' % (li + 1)):
                w1, b1 = fold(blk.conv1, blk.bn1)
                w2, b2 = fold(blk.conv2, blk.bn2)
                w3, b3 = fold(blk.conv3, blk.bn3)
                d = {'w1': rows(w1), 'b1': b1, 'b2': b2, 'w3': rows(w3), 'b3': b3, 'stride': blk.stride, 'mid': w1.shape[0]}
                d['w2'] = rows(w2) if blk.stride != 1 else w2.permute(0, 2, 3, 1).contiguous().to(dtype)      # patch matrix | spatial kernel
                if hasattr(blk, 'downsample'):
                    wd, bd = fold(blk.downsample[0], blk.downsample[1])
                    d['wd'], d['bd'] = rows(wd), bd
                blocks.append(d)
        p['blocks'] = blocks
        p['inner'] = [(rows(m.weight.detach().float()), m.bias.detach().float().contiguous()) for m in fpn.inner_blocks]
        p['layer'] = [(rows(m.weight.detach().float()), m.bias.detach().float().contiguous()) for m in fpn.layer_blocks]
        self._prep = dict(key=key, val=p)
        return p

    def features(self, images, dtype, pyramid=False):
        """images as VGGDetector.features takes them.  Returns (fmap NHWC [B,Hf,Wf,256] in `dtype` -- the 'pool' level --, image
        sizes [(h,w)] after resize, (Hp,Wp) padded size); pyramid=True: a fourth item, the list [P2, P3, P4, P5] (NHWC, strides
        4 .. 32) the detector's RPN and box head read besides the 'pool' level."""
        dev = self.backbone.body.conv1.weight.device
        sizes = [self.transform.resized_hw(*image_hw(im)) for im in images]
        d = self.transform.size_divisible
        Hp = int(math.ceil(max(s[0] for s in sizes) / d) * d)
        Wp = int(math.ceil(max(s[1] for s in sizes) / d) * d)
        B = len(images)
        p = self.prepared(dtype)
        x0 = torch.zeros((B, Hp + 2, Wp + 2, 4), dtype=torch.float32, device=dev)
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
        ops.image_prep_batch(staged, sizes, x0)
        # stem: 7x7 / 2 as patch matrix x [64, 192], ReLU; then MaxPool2d(3, 2, 1)
        cols, H, W = ops.im2col(x0, 7, 2, 3, src_pad=1, C=3, Kp=192, dtype=dtype)
        del x0
        x = ops.gemm(cols, p['stem'][0], p['stem'][1], ops.ACT_RELU).view(B, H, W, 64)
        del cols
        x = ops.maxpool3x3s2(x)
        H, W = x.shape[1], x.shape[2]
        stage_end, done, c_maps = [], 0, []
        for _, _, nblk, _ in LAYERS:
            done += nblk
            stage_end.append(done - 1)
        for bi, blk in enumerate(p['blocks']):
            mid, s = blk['mid'], blk['stride']
            cin = x.shape[3]
            a = ops.gemm(x.view(-1, cin), blk['w1'], blk['b1'], ops.ACT_RELU)                       # 1x1
            ap = torch.zeros((B, H + 2, W + 2, mid), dtype=dtype, device=dev)
            ops.plane_copy(a.view(B, H, W, mid), ap, dst_pad=1)
            if s == 1:
                Ho, Wo = H, W
                bmid = torch.empty((B, H, W, mid), dtype=dtype, device=dev)
                ops.conv3x3_relu(ap, blk['w2'], blk['b2'], bmid, 0)                                 # 3x3 stride 1 (ReLU inside)
                bmid = bmid.view(-1, mid)
            else:
                cols, Ho, Wo = ops.im2col(ap, 3, s, 1, src_pad=1)                                   # 3x3 stride 2 through its patch matrix
                bmid = ops.gemm(cols, blk['w2'], blk['b2'], ops.ACT_RELU)
                del cols
            del a, ap
            y = ops.gemm(bmid, blk['w3'], blk['b3'])                                                # 1x1, no activation before the join
            if 'wd' in blk:
                xs = x if s == 1 else ops.plane_copy(x, torch.empty((B, Ho, Wo, cin), dtype=dtype, device=dev), stride=s)
                idn = ops.gemm(xs.view(-1, cin), blk['wd'], blk['bd'])
            else:
                idn = x.view(-1, cin)
            ops.add_relu_(y, idn)
            H, W = Ho, Wo
            x = y.view(B, H, W, y.shape[1])
            if pyramid and bi in stage_end:
                c_maps.append(x)                                                                    # C2 .. C5
        # FPN level 3 = the top of the pyramid: lateral 1x1, output 3x3 (no activation), then LastLevelMaxPool (kernel 1, stride 2)
        def lateral(c, k):
            return ops.gemm(c.view(-1, c.shape[3]), p['inner'][k][0], p['inner'][k][1]).view(c.shape[0], c.shape[1], c.shape[2], 256)

        def output(inner, k):                       # 3x3, padding 1, no activation: patch matrix x [256, 2304]
            h, w = inner.shape[1], inner.shape[2]
            ip = torch.zeros((B, h + 2, w + 2, 256), dtype=dtype, device=dev)
            ops.plane_copy(inner, ip, dst_pad=1)
            cols, _, _ = ops.im2col(ip, 3, 1, 1, src_pad=1)
            return ops.gemm(cols, p['layer'][k][0], p['layer'][k][1]).view(B, h, w, 256)

        inner = lateral(x, 3)
        p5 = output(inner, 3)
        pool = ops.plane_copy(p5, torch.empty((B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 256), dtype=dtype, device=dev), stride=2)
        if not pyramid:
            return pool, sizes, (Hp, Wp)
        # top-down pathway ([3P] FeaturePyramidNetwork.forward): P_k = output_k(lateral_k(C_k) + upsample(inner_{k+1}))
        levels = [p5]
        for k in (2, 1, 0):
            inner = ops.upsample_add_(lateral(c_maps[k], k), inner)
            levels.insert(0, output(inner, k))
        return pool, sizes, (Hp, Wp), levels
