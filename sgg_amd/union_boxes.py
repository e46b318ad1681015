"""UnionBoxesAndFeats (lib/get_union_boxes.py:16-116), edge_model 'motifs' (default) or 'raw_boxes', on the HIP path.

The reference's conv stack is  Conv(2->d/2,k7,p3) ReLU BN MaxPool(3,2,1) Conv(d/2->d,k3,p1) ReLU BN  with BOTH convs
at stride 16 (the `stide` typo, :40-43), so on the 27x27 raster it produces [E,d,1,1], broadcast-added over the 7x7
union features (:101).  Here: raster patches (HIP) -> GEMM+ReLU+BN (MFMA) -> max over the 2x2 map (HIP) ->
centre-tap GEMM+ReLU+BN (MFMA) = rect_feat[E,d]; the broadcast add is folded into fc6 by linearity (see
rel_model_stanford.py) or fused into RoIAlign / applied by sgg_bcast_add when the raw sum is requested.
"""
import torch
import torch.nn as nn

from . import ops

BATCHNORM_MOMENTUM = 0.01  # config.py:34


def _pad_k(w, mult=64):
    k = w.shape[1]
    kp = (k + mult - 1) // mult * mult
    if kp == k:
        return w.contiguous()
    out = torch.zeros((w.shape[0], kp), dtype=w.dtype, device=w.device)
    out[:, :k] = w
    return out


def fold_rect_conv(p, dtype, eps=1e-5):
    """state-dict tensors ('conv.0.weight', ..., 'conv.6.running_var') -> GEMM operands with eval-mode BN folded
    into a per-channel post scale/shift (y = relu(conv)+b -> *s + t)."""
    f = lambda k: p[k].detach().float()
    d2, d = p['conv.0.weight'].shape[0], p['conv.4.weight'].shape[0]
    s1 = f('conv.2.weight') / torch.sqrt(f('conv.2.running_var') + eps)
    t1 = f('conv.2.bias') - f('conv.2.running_mean') * s1
    s2 = f('conv.6.weight') / torch.sqrt(f('conv.6.running_var') + eps)
    t2 = f('conv.6.bias') - f('conv.6.running_mean') * s2
    w1 = _pad_k(f('conv.0.weight').reshape(d2, 98), 128)              # k = c*49 + ky*7 + kx
    w2 = _pad_k(f('conv.4.weight')[:, :, 1, 1].contiguous(), 64)     # only the centre tap sees data (1x1 map, p1)
    return dict(w1=w1.to(dtype).contiguous(), b1=f('conv.0.bias').contiguous(), s1=s1.contiguous(), t1=t1.contiguous(),
                w2=w2.to(dtype).contiguous(), b2=f('conv.4.bias').contiguous(), s2=s2.contiguous(), t2=t2.contiguous(),
                d2=d2, d=d, k2=w2.shape[1])


def rect_feat(rois, pairs, prep, dtype, P=27, im_sizes=None):
    """rois f32[N,5], pairs i64[E,2] -> conv(rects)[E,d] in `dtype` (eval-mode BN).  im_sizes: given for edge_model 'raw_boxes'."""
    E = pairs.shape[0]
    patches = ops.union_rect_patches(rois, pairs, dtype, P, prep['w1'].shape[1], im_sizes=im_sizes)   # [4E,128]
    h1 = torch.zeros((4 * E, prep['k2']), dtype=dtype, device=rois.device) if prep['k2'] != prep['d2'] else \
        torch.empty((4 * E, prep['k2']), dtype=dtype, device=rois.device)
    ops.gemm(patches, prep['w1'], prep['b1'], ops.ACT_RELU, dtype, post_scale=prep['s1'], post_shift=prep['t1'], out=h1)
    h2 = ops.max4_rows(h1)                                                                    # [E,k2]
    return ops.gemm(h2, prep['w2'], prep['b2'], ops.ACT_RELU, dtype, post_scale=prep['s2'], post_shift=prep['t2'])


class UnionBoxesAndFeats(nn.Module):
    """Same constructor, parameter names (conv.{0,2,4,6}.*) and forward signature as the reference module."""

    def __init__(self, edge_model='motifs', pooling_size=7, stride=16, dim=256, concat=False, use_feats=True):
        super(UnionBoxesAndFeats, self).__init__()
        if edge_model not in ('motifs', 'raw_boxes'):     # lib/get_union_boxes.py:26-38
            raise NotImplementedError(edge_model)
        if concat:
            raise NotImplementedError('concat=True')
        self.edge_model, self.pooling_size, self.stride, self.dim = edge_model, pooling_size, stride, dim
        self.use_feats, self.concat = use_feats, concat
        # every conv gets the enclosing `stride` (lib/get_union_boxes.py:40-43)
        self.conv = nn.Sequential(
            nn.Conv2d(2, dim // 2, kernel_size=7, stride=stride, padding=3, bias=True),
            nn.ReLU(inplace=True),
            nn.BatchNorm2d(dim // 2, momentum=BATCHNORM_MOMENTUM),
            nn.MaxPool2d(kernel_size=3, stride=2, padding=1),
            nn.Conv2d(dim // 2, dim, kernel_size=3, stride=stride, padding=1, bias=True),
            nn.ReLU(inplace=True),
            nn.BatchNorm2d(dim, momentum=BATCHNORM_MOMENTUM),
        )
        self._prep = {}
        # train-mode forwards of the HIP path count here; the two BatchNorms' `num_batches_tracked` buffers (device scalars that only
        # checkpoints read: momentum is a constant) receive the count when a state_dict is taken or flush_batch_counts() is called --
        # not with two device launches per step
        self._pending_batches = 0
        self.register_state_dict_pre_hook(lambda module, prefix, keep_vars: module.flush_batch_counts())

    def count_train_batch(self):
        self._pending_batches += 1

    def flush_batch_counts(self):
        n, self._pending_batches = self._pending_batches, 0
        if n:
            with torch.no_grad():
                self.conv[2].num_batches_tracked += n
                self.conv[6].num_batches_tracked += n

    def prepared(self, dtype):
        key = (dtype, tuple(p._version for p in self.parameters()), tuple(b._version for b in self.buffers()),
               next(self.parameters()).device)
        if self._prep.get('key') != key:
            sd = {k: v for k, v in self.state_dict().items()}
            self._prep = dict(key=key, val=fold_rect_conv(sd, dtype, self.conv[2].eps))
        return self._prep['val']

    def raster_sizes(self, im_sizes):
        """what the raster kernels need besides the boxes: nothing for 'motifs', the image sizes for 'raw_boxes' (:71-78)"""
        if self.edge_model == 'motifs':
            return None
        if im_sizes is None:
            raise ValueError("edge_model 'raw_boxes' needs im_sizes")
        return im_sizes

    def rect_feat(self, rois, union_inds, dtype, im_sizes=None):
        if self.training:
            raise NotImplementedError('train-mode BatchNorm statistics of the rect conv are not on the HIP path yet')
        return rect_feat(rois.float().contiguous(), union_inds.contiguous(), self.prepared(dtype), dtype,
                         self.pooling_size * 4 - 1, im_sizes=self.raster_sizes(im_sizes))

    def forward(self, union_pools, rois, union_inds, im_sizes=None):
        """union_pools [E,dim,7,7] -> union_pools + conv(rects) (lib/get_union_boxes.py:101), a new tensor."""
        E, C = union_pools.shape[0], union_pools.shape[1]
        x = union_pools
        if x.dtype not in (torch.float32, torch.bfloat16, torch.float16):
            x = x.float()
        rf = self.rect_feat(rois, union_inds, x.dtype, im_sizes).float().contiguous()
        x = x.contiguous().clone()
        ops.bcast_add_(x.view(E, C, -1), rf)
        return x
