"""The GAN feature-augmentation model of BASELINE config 5 behind the reference's interface: `GAN`
(augment/gan.py:17-259: constructor arguments, parameter names -- the reference's `state_dict()` loads by name -- `forward`,
`loss`, `loss_fn`), `dummy_nodes` (augment/gan.py:262-289), `RefinementNetwork` / `RefinementModule` (augment/crn.py:64-142).

What runs where.  Everything with a contraction or a gather in it is this package's HIP code: the scene-graph convolutions gather /
pool through `triple_gather` / `triple_pool` and multiply through `sgg_amd.dense.Linear`; the 3x3 / 1x1 convolutions on the 7x7
patches, the cascaded refinement network on the 38x38 canvas and the three spectral-norm discriminators are `sgg_amd.dense.Conv2d`
(patch matrix + the exact-fp32 MFMA GEMM, backward through the same GEMM and `sgg_col2im`); the per-object patches are resampled onto
the canvas and summed per image by the one-launch `boxes_to_layout` kernel (sgg_amd/gan_ops.py, csrc/gan.hip) -- each with its adjoint,
so the whole model trains.  The networks are channels-last from end to end: a feature map is the [B*H*W, C] row matrix the GEMM reads,
`forward` returns an NCHW *view* of it (what the SGG model's RoIAlign takes without a copy).  Elementwise pieces (activations, batch
normalisation, 2x2 average pools, nearest-neighbour upsampling, the losses) are torch expressions.  `features.hdf5` (the `vis_cond`
file of real per-class 512x7x7 features) is read by sgg_amd/hdf5_lite.py.

Not carried: the GloVe tables (`lib/word_vectors.py`; `embed_objs` is only used by the scene-graph perturbations
(sgg_amd/sg_perturb.py), and `init_embed` "led to worse results" and is off in the reference): pass `embed_objs=` / `embed_rels=`
tensors if you have them, otherwise the attributes stay None and `init_embed=True` raises.
"""
import numpy as np
import torch
import torch.nn as nn
from torch.nn import functional as F

from . import dense
from .gan_ops import GraphTripleConvNet, boxes_to_layout_nhwc


def to_nhwc(x):
    """NCHW tensor -> channels-last tensor [B,H,W,C] (no copy when x is an NCHW view of channels-last memory, as the detector's maps are)"""
    return x.permute(0, 2, 3, 1)


def to_nchw(x):
    return x.permute(0, 3, 1, 2)


class _OnPlanes(nn.Module):
    """runs a torch NCHW spatial op on a channels-last tensor through views (the op sees channels_last memory)"""

    def __init__(self, fn):
        super(_OnPlanes, self).__init__()
        self.fn = fn

    def forward(self, x):
        return to_nhwc(self.fn(to_nchw(x)))


class InstanceNormRows(nn.Module):
    """nn.InstanceNorm2d (no affine, no running statistics: nothing in the state_dict) on channels-last maps"""

    def __init__(self, eps=1e-5):
        super(InstanceNormRows, self).__init__()
        self.eps = eps

    def forward(self, x):
        mean = x.mean((1, 2), keepdim=True)
        var = (x - mean).square().mean((1, 2), keepdim=True)
        return (x - mean) * torch.rsqrt(var + self.eps)


# ------------------------------------------------------------------------------------------------ augment/crn.py
def get_normalization_2d(channels, normalization):
    """augment/crn.py:38-46, channels-last layers"""
    table = {'instance': lambda: InstanceNormRows(), 'batch': lambda: dense.BatchNormRows(channels), 'none': lambda: None}
    if normalization not in table:
        raise ValueError('Unrecognized normalization type "%s"' % normalization)
    return table[normalization]()


def get_activation(name):
    """augment/crn.py:49-62: 'relu', 'leakyrelu', 'leakyrelu-<slope>' -- and, as there, EVERY name ends up a LeakyReLU (the
    reference overwrites `name` before the lookup); the slope defaults to torch's 0.01."""
    slope = float(name.split('-')[1]) if name.lower().startswith('leakyrelu') and '-' in name else 0.01
    return nn.LeakyReLU(negative_slope=slope)


class RefinementModule(nn.Module):
    """One scale of the cascade (augment/crn.py:64-94): [layout at this scale | incoming features] -> two 3x3 convolutions, each followed
    by normalisation and activation.  `net` keeps the reference's Sequential indices (a missing normalisation shifts them, as there).
    Channels-last: layout [B,h,w,Dl], feats [B,h,w,Df] -> [B,h,w,Dout]."""

    def __init__(self, layout_dim, input_dim, output_dim, normalization='instance', activation='leakyrelu'):
        super(RefinementModule, self).__init__()
        stack = []
        for n_in in (layout_dim + input_dim, output_dim):
            conv = dense.Conv2d(n_in, output_dim, 3, padding=1)
            nn.init.kaiming_normal_(conv.weight)
            stack += [conv, get_normalization_2d(output_dim, normalization), get_activation(activation)]
        self.net = nn.Sequential(*[m for m in stack if m is not None])

    def forward(self, layout, feats):
        assert layout.shape[1:3] == feats.shape[1:3], (tuple(layout.shape), tuple(feats.shape))
        return self.net(torch.cat((layout, feats), dim=-1))


class RefinementNetwork(nn.Module):
    """Cascaded refinement (augment/crn.py:97-142) over len(dims) - 1 scales.  The coarsest scale is the layout size halved once per
    module; every module sees the layout average-pooled to its scale and the previous module's output upsampled (nearest) to it -- by
    2 between modules, and straight to the layout's own size for the last one (three modules on a 38x38 layout: 4 -> 8 -> 16 -> 38).  The first
    module's incoming features are one zero channel.  A 3x3 convolution ends it.  forward: NCHW in, NCHW (view) out."""

    def __init__(self, dims, normalization='instance', activation='leakyrelu'):
        super(RefinementNetwork, self).__init__()
        self.refinement_modules = nn.ModuleList(
            RefinementModule(dims[0], dims[i - 1] if i > 1 else 1, dims[i], normalization=normalization, activation=activation)
            for i in range(1, len(dims)))
        last = dense.Conv2d(dims[-1], dims[-1], 3, padding=1)
        nn.init.kaiming_normal_(last.weight)
        self.output_conv = nn.Sequential(last)

    def scales(self, H, W):
        n = len(self.refinement_modules)
        h, w = H >> n, W >> n
        assert h > 0 and w > 0, 'the layout is too small for %d refinement modules' % n
        return [(h << (i + 1), w << (i + 1)) for i in range(n - 1)] + [(H, W)]

    def refine(self, layout):
        """channels-last: [B,H,W,D0] -> [B,H,W,dims[-1]]"""
        B, H, W, _ = layout.shape
        planes = to_nchw(layout)
        feats = None
        for mod, (h, w) in zip(self.refinement_modules, self.scales(H, W)):
            at_scale = layout if (h, w) == (H, W) else to_nhwc(F.adaptive_avg_pool2d(planes, (h, w)))
            if feats is None:
                feats = layout.new_zeros((B, h, w, 1))
            else:
                feats = to_nhwc(F.interpolate(to_nchw(feats), size=(h, w), mode='nearest'))
            feats = mod(at_scale, feats)
        return self.output_conv(feats)

    def forward(self, layout):
        self.layout = layout
        return to_nchw(self.refine(to_nhwc(layout)))


# ------------------------------------------------------------------------------------------------ augment/gan.py
def dummy_nodes(gt_objs, gt_boxes, gt_rels):
    """augment/gan.py:262-289: one dummy node (class 0, box [0,0,1,1]) per image, connected both ways to every object of the image
    with predicate 0; node indices become global over the batch (dummies included).  gt_objs i64[N,2] (image, class), gt_boxes
    [N,4], gt_rels i64[T,4] (image, subject, object, predicate) with image-LOCAL subject / object indices, both sorted by image.
    No per-object loop: segment offsets and index arithmetic on the device the inputs live on."""
    dev = gt_objs.device
    im_o, im_r = gt_objs[:, 0], gt_rels[:, 0]
    B = int(im_o.max().item()) + 1 if gt_objs.shape[0] else 0
    n = torch.bincount(im_o, minlength=B)                                   # objects per image
    t = torch.bincount(im_r, minlength=B)                                   # relations per image
    if B == 0 or int((n == 0).sum()) or int((t == 0).sum()) or im_r.shape[0] == 0 or int(im_r.max()) >= B:
        raise IndexError('dummy_nodes: every image 0..B-1 needs at least one object and one relation (the reference indexes its '
                         'per-image relation list by image id)')
    N = gt_objs.shape[0]
    off = torch.cumsum(n + 1, 0) - (n + 1)                                  # first new node id of every image
    # nodes: image i's objects, then its dummy
    new_pos = torch.arange(N, device=dev) + im_o                           # an object moves down by the dummies of earlier images
    dummy_pos = off + n
    objs = gt_objs.new_zeros((N + B, 2))
    objs[new_pos] = gt_objs
    objs[dummy_pos, 0] = torch.arange(B, device=dev, dtype=gt_objs.dtype)
    boxes = gt_boxes.new_zeros((N + B, 4))
    boxes[new_pos] = gt_boxes
    boxes[dummy_pos] = torch.tensor([0, 0, 1, 1], dtype=gt_boxes.dtype, device=dev)
    # relations: image i's own (indices + off[i]), then dummy -> object k, then object k -> dummy
    T = gt_rels.shape[0]
    r_off = torch.cumsum(t + 2 * n, 0) - (t + 2 * n)                        # first new relation row of every image
    t_start = torch.cumsum(t, 0) - t
    rels = gt_rels.new_zeros((T + 2 * N, 4))
    own = r_off[im_r] + (torch.arange(T, device=dev) - t_start[im_r])
    rels[own] = gt_rels
    rels[own, 1] += off[im_r]
    rels[own, 2] += off[im_r]
    k = torch.arange(N, device=dev) - (torch.cumsum(n, 0) - n)[im_o]        # local object index
    out_rows = r_off[im_o] + t[im_o] + k
    in_rows = out_rows + n[im_o]
    dummy_of = (off + n)[im_o]
    rels[out_rows, 0] = im_o
    rels[out_rows, 1] = dummy_of
    rels[out_rows, 2] = off[im_o] + k
    rels[in_rows, 0] = im_o
    rels[in_rows, 1] = off[im_o] + k
    rels[in_rows, 2] = dummy_of
    return objs, boxes, rels


class GAN(nn.Module):
    """augment/gan.py:17-259.  Generator: class / predicate embeddings -> scene-graph convolutions (G_gcn) -> per-object 32x7x7
    features -> G_node (two 3x3 convs) [-> concatenated with real 512x7x7 features of the same class when `vis_cond` names a
    features.hdf5] -> G_proj (1x1) -> boxes_to_layout onto the fmap_sz^2 canvas -> cascaded refinement -> ReLU = a fake
    512 x fmap_sz x fmap_sz detector feature map.  Discriminators: D_nodes / D_edges on class-conditioned 7x7 RoI features,
    D_global on the feature map."""

    def __init__(self, obj_classes, rel_classes, embed_dim=200, hidden_dim=64, n_ch=512, pool_sz=7, fmap_sz=38,
                 losses=('D', 'G', 'rec'), SN=True, BN=True, n_layers_G=5, vis_cond=None, init_embed=False, largeD=False,
                 data_dir='', device='cuda', embed_objs=None, embed_rels=None):
        super(GAN, self).__init__()
        self.obj_classes, self.rel_classes = obj_classes, rel_classes
        self.embed_dim, self.n_ch, self.pool_sz, self.fmap_sz = embed_dim, n_ch, pool_sz, fmap_sz
        self.obj_dim = pool_sz ** 2 * n_ch
        self.losses, self.SN, self.BN, self.vis_cond, self.largeD, self.device = losses, SN, BN, vis_cond, largeD, device
        self.h5_data = None
        if vis_cond is not None:
            from .hdf5_lite import File
            self.h5_data = File(vis_cond)

        self.G_obj_embed = nn.Embedding(len(obj_classes), embed_dim)
        self.G_rel_embed = nn.Embedding(len(rel_classes), embed_dim)

        # Discriminators as layer tables: (out channels, kernel) per convolution, all without padding; 'A' = LeakyReLU(0.2),
        # 'R' = ReLU, 'P' / 'Pc' = AvgPool2d(2) / with ceil_mode.  Positions that the reference fills with nn.Identity() when an
        # option is off keep an Identity here too: the Sequential indices are part of the state_dict keys.
        def stack(n_in, table):
            mods = []
            for item in table:
                if item is None:
                    mods.append(nn.Identity())
                elif item == 'A':
                    mods.append(nn.LeakyReLU(0.2))
                elif item == 'R':
                    mods.append(nn.ReLU())
                elif item in ('P', 'Pc'):
                    mods.append(_OnPlanes(nn.AvgPool2d(2, ceil_mode=item == 'Pc')))
                else:
                    n_out, ks = item
                    mods.append(dense.Conv2d(n_in, n_out, ks, padding=0, spectral=SN))
                    n_in = n_out
            return nn.Sequential(*(mods + [nn.Flatten()]))

        # RoI features (n_ch + classes) x 7 x 7 -> 5x5 -> 3x3 -> 3x3 -> one logit
        roi_table = [(n_ch // 2, 3), 'R', (n_ch // 4, 3), 'R', (n_ch // 8, 1), 'R', (1, 3)]
        self.D_nodes = stack(n_ch + len(obj_classes), roi_table)
        self.D_edges = stack(n_ch + len(rel_classes), roi_table)
        # feature map n_ch x 38 x 38 -> 36 -> (pool) 18 -> 16 -> (pool) 8 -> 6 -> (pool) 3 -> one logit; largeD adds a 1x1 layer per scale
        wide = lambda c: [(c, 1), 'A'] if largeD else [None, None]                                       # noqa: E731
        self.D_global = stack(n_ch, [(n_ch // 2, 3), 'A'] + wide(n_ch // 2) + ['Pc' if fmap_sz > 24 else None] +
                              [(n_ch // 2, 3), 'A'] + wide(n_ch // 2) + ['P'] +
                              [(n_ch // 4, 3), 'A'] + wide(n_ch // 4) + ['P', (1, 3)])

        self.G_gcn = GraphTripleConvNet(input_dim=embed_dim + 4, input_edge_dim=embed_dim,
                                        output_dim=hidden_dim // 2 * pool_sz * pool_sz, num_layers=n_layers_G,
                                        hidden_dim=hidden_dim, pooling='avg', mlp_normalization='batch' if BN else 'none')
        self.G_node = nn.Sequential(dense.Conv2d(hidden_dim // 2, hidden_dim, 3, padding=1), nn.ReLU(),
                                    dense.Conv2d(hidden_dim, hidden_dim, 3, padding=1), nn.ReLU())
        self.G_proj = dense.Conv2d(hidden_dim + int(vis_cond is not None) * n_ch, hidden_dim, 1)
        self.G_refine = RefinementNetwork(dims=(hidden_dim, n_ch // 4, n_ch // 2, n_ch), normalization='batch',
                                          activation='leakyrelu-0.2')

        norm = lambda e: None if e is None else (e / torch.norm(e, 2, dim=1, keepdim=True)).to(device)   # noqa: E731
        self.embed_objs, self.embed_rels = norm(embed_objs), norm(embed_rels)
        if init_embed:
            if self.embed_objs is None or self.embed_rels is None:
                raise ValueError('GAN(init_embed=True) needs embed_objs= / embed_rels= (the GloVe tables are not carried)')
            assert self.G_obj_embed.weight.shape == self.embed_objs.shape and self.G_rel_embed.weight.shape == self.embed_rels.shape
            self.G_obj_embed.weight.data = self.embed_objs.clone()
            self.G_rel_embed.weight.data = self.embed_rels.clone()

    # targets of the discriminator losses: made at the size asked for (the reference slices two preallocated 50000-row vectors)
    def y_real(self, n):
        return torch.ones((n, 1), device=self.device)

    def y_fake(self, n):
        return torch.zeros((n, 1), device=self.device)

    def loss_fn(self, predictions, is_fake=True, updateD=False):
        """augment/gan.py:162-171: binary cross-entropy of the logits against "fake" / "real" for a discriminator update, against
        "real" for the generator's (which only ever scores generated samples)."""
        n = len(predictions)
        if not updateD:
            assert is_fake
            return F.binary_cross_entropy_with_logits(predictions, self.y_real(n))
        return F.binary_cross_entropy_with_logits(predictions, self.y_fake(n) if is_fake else self.y_real(n))

    def sample_real_features(self, classes):
        """augment/gan.py:193-199: one random real 512x7x7 feature of each object's class out of features.hdf5 (numpy's global RNG,
        one `permutation(n)[0]` draw per object, as there)."""
        feats = []
        for cls in classes.tolist():
            assert cls > 0, 'background objects are not expected here'
            dset = self.h5_data[self.obj_classes[cls]]
            ind = int(np.random.permutation(dset.shape[0])[0])
            feats.append(torch.from_numpy(np.asarray(dset[ind])).view(1, self.n_ch, self.pool_sz, self.pool_sz))
        return torch.cat(feats)

    def generate(self, gt_objects, boxes_scaled, gt_rels):
        """The generator, channels-last: -> fake feature maps [B, fmap_sz, fmap_sz, n_ch] (augment/gan.py:174-208)."""
        P = self.pool_sz
        objs, boxes, rels = dummy_nodes(gt_objects, boxes_scaled, gt_rels)        # one all-connected extra node per image
        node_in = torch.cat((self.G_obj_embed(objs[:, -1]), boxes), dim=1)        # class embedding | box
        node_out, _ = self.G_gcn(node_in, self.G_rel_embed(rels[:, -1]), rels[:, 1:3])
        real = torch.nonzero(objs[:, -1]).view(-1)                                # the dummies (class 0) leave again
        objs, boxes = objs[real], boxes[real]
        n = real.numel()
        # a node's vector is a (channels, P, P) block in the reference's order; the convolutions here want (P, P, channels)
        patches = self.G_node(node_out[real].view(n, -1, P, P).permute(0, 2, 3, 1))
        assert patches.shape[0] == objs.shape[0] == boxes.shape[0], (patches.shape, objs.shape, boxes.shape)
        if self.h5_data is not None:       # conditioning: a real feature of the same class in front of the generated one
            seen = self.sample_real_features(objs[:, -1].detach().cpu()).to(patches)
            patches = torch.cat((seen.permute(0, 2, 3, 1), patches), dim=-1)
        canvas = boxes_to_layout_nhwc(self.G_proj(patches), boxes, objs[:, 0], self.fmap_sz, self.fmap_sz, pooling='sum')
        return F.relu(self.G_refine.refine(canvas))

    def forward(self, gt_objects, boxes_scaled, gt_rels):
        """augment/gan.py:174-208 -> fake feature maps [B, n_ch, fmap_sz, fmap_sz] (an NCHW view of channels-last memory, the form the
        detector's own maps have in this package)"""
        return to_nchw(self.generate(gt_objects, boxes_scaled, gt_rels))

    def _roi_planes(self, feats, labels, n_classes):
        """RoI features [n, n_ch*P*P] or [n, n_ch, P, P] (channel-major, as RoIAlign writes them) -> [n, P, P, n_ch + n_classes]: the
        discriminator's channels-last input with the class appended as one-hot planes (augment/gan.py:222-231)."""
        P = self.pool_sz
        n = feats.shape[0]
        x = feats.reshape(n, -1, P, P).permute(0, 2, 3, 1)
        hot = F.one_hot(labels.view(-1), n_classes).to(x.dtype)
        return torch.cat((x, hot[:, None, None, :].expand(n, P, P, n_classes)), dim=-1)

    def _score_rois(self, net, feats, labels, n_classes):
        """net(_roi_planes(feats, labels)) without forming the one-hot planes (augment/gan.py:222-231 concatenates n_classes constant planes
        to every RoI feature: 1 704 of 1 960 input channels of D_nodes with GQA's vocabulary, 311 of 567 of D_edges).  The first layer is
        a 3x3 convolution WITHOUT padding, so each of its outputs sees all nine taps of a plane that is constant over the patch: the class
        channels contribute sum_taps W[:, n_ch + label, ky, kx] -- one row of a [n_classes, Cout] table, the same at every output position.
        conv([x | onehot]) = conv(x; W[:, :n_ch]) + table[label]: the same sum in another order, 2.2x (edges) / 7.7x (nodes) fewer
        multiply-adds and patch-matrix bytes; the gradient reaches W's class channels through the table (index_select's adjoint)."""
        P = self.pool_sz
        n = feats.shape[0]
        x = feats.reshape(n, -1, P, P).permute(0, 2, 3, 1)
        first = net[0]
        assert first.padding == 0 and first.in_channels == x.shape[-1] + n_classes, (first.padding, first.in_channels, x.shape, n_classes)
        w = first.effective_weight()
        C = x.shape[-1]
        y = dense.conv2d(x, w[:, :C], first.bias, padding=0)
        table = w[:, C:].sum((2, 3)).t()                                 # [n_classes, Cout]
        y = y + table.index_select(0, labels.view(-1))[:, None, None, :]
        for m in list(net)[1:]:
            y = m(y)
        return y

    def loss(self, features_real=None, features_fake=None, is_nodes=False, updateD=False, labels_fake=None, labels_real=None,
             is_fmaps=False):
        """augment/gan.py:211-259 -> {'D_obj' | 'D_rel' | 'D_fmap' | 'G_obj' | 'G_rel' | 'G_fmap': loss} (or {} when the side is off).
        A discriminator update scores real and generated samples (both detached); a generator update scores the generated ones only,
        with the gradient flowing back into them."""
        side = 'D' if updateD else 'G'
        if side not in self.losses:
            return {}
        kind = 'fmap' if is_fmaps else ('obj' if is_nodes else 'rel')
        net = {'fmap': self.D_global, 'obj': self.D_nodes, 'rel': self.D_edges}[kind]
        if not updateD:
            assert labels_real is None and features_real is None, 'do not need real labels/features in case of G update'
        # the SGG model may run in a 16-bit mode (its RoI features and feature maps then arrive in it); the discriminators are fp32 layers
        f32 = lambda t: None if t is None else t.to(torch.float32)        # noqa: E731
        fake, real = f32(features_fake), f32(features_real)
        if is_fmaps:
            score = lambda t, _labels: net(to_nhwc(t))                    # noqa: E731
        else:
            n_classes = len(self.obj_classes if is_nodes else self.rel_classes)
            score = lambda t, labels: self._score_rois(net, t, labels, n_classes)   # noqa: E731
        if updateD:
            total = self.loss_fn(score(real.detach(), labels_fake if labels_real is None else labels_real), is_fake=False, updateD=True) + \
                self.loss_fn(score(fake.detach(), labels_fake), is_fake=True, updateD=True)
        else:
            total = self.loss_fn(score(fake, labels_fake), is_fake=True, updateD=False)
        return {side + '_' + kind: total}


# ------------------------------------------------------------------------------------------------ main.py:124-194
def _dp_world():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def allreduce_grads_(params, scale=1.0):
    """SUM the gradients of `params` over the ranks in place (one flat fp32 message: the GAN's ~20 M parameters), then multiply by `scale`.
    Parameters without a gradient on this rank contribute zeros (and receive the others' sum)."""
    import torch.distributed as dist
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    if _dp_world() > 1:
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).float() for p in params])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if scale != 1.0:
            flat.mul_(scale)
        off = 0
        for p in params:
            n = p.numel()
            g = flat[off:off + n].view_as(p).to(p.dtype)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
    elif scale != 1.0:
        for p in params:
            if p.grad is not None:
                p.grad.mul_(scale)


def gan_train_step(sgg_model, gan, res, gt_boxes, gt_objects, gt_rels, optimizer, G_optimizer, D_optimizer, *, ganw=1.0,
                   attachG=False, ganlosses=('D', 'G', 'rec'), loss_type='baseline', loss_weights=(1, 1, 1), clip=5.0,
                   gt_objects_fake=None, trainer=None):
    """The GAN part of one training iteration -- what main.py:124-194 does after the SGG model's own update, with the same calls in
    the same order: generate feature maps from the (perturbed) scene graph, extract node / edge features from them with the SGG
    model's RoIAlign (differentiable into the maps: sgg_roi_align_bwd), classify them with the SGG head, update G (adversarial +
    reconstruction losses; the SGG model too when 'rec' is on), then update D on real vs generated features.
    `res` is the Result of the SGG model's training forward on the same batch.  -> {loss name: value} as main.py logs them.
    gt_objects_fake: the perturbed objects, `sgg_amd.sg_perturb.SceneGraphPerturb(...).perturb(gt_objects.clone(), gt_rels.clone())`
    (main.py:131-134); default = the real ones (`-perturb` off).

    Data parallel (one process per GPU, torch.distributed initialised; BASELINE configs[4]): every rank runs this on ITS images.  The
    adversarial losses are means over a rank's nodes / edges / images: each is weighted by (local rows / global rows) and the
    gradients of G, D and the SGG model are SUM-reduced over the ranks before their optimiser steps, so that one iteration equals
    the single-process iteration on the concatenated batch (tests/test_gan_dist_gpu.py).  The reconstruction losses use the trainer's
    global normalisers.  `trainer` (a sgg_amd.trainer.Trainer of `sgg_model`): the SGG model's update then goes through
    Trainer.update() -- the backward's early-started reduce-scatters / all-reduces, wire-dtype buffers, sharded fused clip + SGD,
    loss scale of the f16 mode -- instead of `optimizer.step()` (pass optimizer=None).  Without a trainer the gradients are
    all-reduced here and `optimizer` steps on them.  The GAN's own BatchNorm layers (BatchNorm2d of the refinement network,
    BatchNorm1d of the graph convolutions when BN=True) use batch statistics: convert them with
    `sgg_amd.dense.sync_batchnorm_(gan)` (the channels-last counterpart of torch.nn.SyncBatchNorm.convert_sync_batchnorm) for
    statistics over every rank's images (what the test does), or keep replica-local statistics."""
    from .trainer import Trainer
    import torch.distributed as dist
    world = _dp_world()
    gan.train()
    if gt_objects_fake is None:
        gt_objects_fake = gt_objects.clone()
    fmaps = gan(gt_objects_fake, sgg_model.get_scaled_boxes(gt_boxes, res.im_inds, res.im_sizes_org), gt_rels)
    # a data-parallel Trainer leaves its gradient hook on the model: without `trainer` every gradient of this function's backward
    # must land in p.grad (the plain optimiser reads it there)
    saved_hook = getattr(sgg_model, '_grad_ready_hook', None)
    if trainer is None and saved_hook is not None:
        sgg_model._grad_ready_hook = None
    try:
        nodes_fake, edges_fake = sgg_model.node_edge_features(fmaps, res.rois, res.rel_inds[:, 1:], res.im_sizes)
        obj_fake, rel_fake = sgg_model.predict(nodes_fake if attachG else nodes_fake.detach(), edges_fake if attachG else edges_fake.detach(),
                                               res.rel_inds, rois=res.rois, im_sizes=res.im_sizes)
        # rows behind the three kinds of discriminator means, on this rank and over all ranks
        rows = torch.tensor([float(nodes_fake.shape[0]), float(edges_fake.shape[0]), float(fmaps.shape[0])], dtype=torch.float64, device=fmaps.device)
        share = {'obj': 1.0, 'rel': 1.0, 'fmap': 1.0}
        if world > 1:
            tot = rows.clone()
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            share = {k: float(rows[i] / tot[i]) for i, k in enumerate(('obj', 'rel', 'fmap'))}
        weigh = lambda d: {k: ganw * share[k.split('_', 1)[1]] * v for k, v in d.items()}
        G_params = [p for n, p in gan.named_parameters() if n.startswith('G_')]
        D_params = [p for n, p in gan.named_parameters() if n.startswith('D_')]
        scale = trainer.loss_scale if trainer is not None else 1.0        # f16 compute of the SGG head: scaled backward, unscaled updates
        # ---- generator (main.py:151-176)
        if trainer is not None:
            trainer.opt.zero_grad()
        else:
            optimizer.zero_grad()
        G_optimizer.zero_grad()
        losses = {}
        losses_G = {}
        # The generator's update scores generated samples with the discriminators but never steps them: main.py:151-176 lets autograd fill the
        # discriminators' .grad all the same and D_optimizer.zero_grad() (:179) throws it away.  The same update without that work: the
        # discriminators' parameters do not require a gradient while the generator's losses are built and back-propagated (their weight-gradient
        # contractions -- the largest of the iteration: D_edges' first layer alone is a [256 x 198400] . [198400 x 2304] product at the GQA
        # configuration -- are never formed); the gradient THROUGH them into the generated features is untouched.
        d_req = [(p, p.requires_grad) for p in D_params]
        for p, _ in d_req:
            p.requires_grad_(False)
        losses_G.update(gan.loss(features_fake=nodes_fake, is_nodes=True, labels_fake=gt_objects_fake[:, -1]))
        losses_G.update(gan.loss(features_fake=edges_fake, labels_fake=res.rel_labels[:, -1]))
        losses_G.update(gan.loss(features_fake=fmaps, is_fmaps=True))
        losses_G = weigh(losses_G)
        if 'rec' in ganlosses:
            class _R(object):        # the loss code of the trainer takes a Result-shaped object
                pass
            r = _R()
            r.rm_obj_dists, r.rm_obj_labels, r.rel_dists, r.rel_labels = obj_fake, gt_objects_fake[:, -1], rel_fake, res.rel_labels
            rec = trainer.losses(r) if trainer is not None else Trainer.losses(_LossCfg(loss_type, loss_weights, world), r)
            losses_G['rec'] = rec                                    # node + edge reconstruction losses (main.py:163-170), summed
        if losses_G:
            total = sum(losses_G.values())
            sgg_model._loss_scaled = trainer is not None
            try:
                (total * scale if scale != 1.0 else total).backward()
            finally:
                sgg_model._loss_scaled = False
            if 'rec' in ganlosses:
                if trainer is not None:
                    trainer.update()                                 # reduce over ranks + clip + SGD (sharded / fused), unscales
                else:
                    allreduce_grads_([p for p in sgg_model.parameters() if p.requires_grad])
                    if clip:
                        torch.nn.utils.clip_grad_norm_([p for p in sgg_model.parameters() if p.grad is not None], clip)
                    optimizer.step()
            allreduce_grads_(G_params, 1.0 / scale)
            # a scaled f16 backward that overflowed: the SGG update above was skipped by its norm check; the generator's optimiser has no
            # such guard, so an inf / NaN gradient would go into G's weights (and Adam's moments) for good -- skip its step too
            g_ok = True
            if scale != 1.0:
                gs = [p.grad for p in G_params if p.grad is not None]
                g_ok = bool(torch.stack([g.isfinite().all() for g in gs]).all()) if gs else True
            if g_ok:
                G_optimizer.step()
            losses.update(losses_G)
        for p_, req_ in d_req:                   # (the discriminators learn again)
            p_.requires_grad_(req_)
        d_req = []
        # ---- discriminators (main.py:178-191)
        D_optimizer.zero_grad()
        losses_D = {}
        losses_D.update(gan.loss(res.node_feat, nodes_fake, is_nodes=True, updateD=True, labels_fake=gt_objects_fake[:, -1], labels_real=gt_objects[:, -1]))
        losses_D.update(gan.loss(res.edge_feat, edges_fake, updateD=True, labels_fake=res.rel_labels[:, -1]))
        losses_D.update(gan.loss(res.fmap, fmaps, updateD=True, is_fmaps=True))
        losses_D = weigh(losses_D)
        if losses_D:
            sum(losses_D.values()).backward()
            allreduce_grads_(D_params)
            D_optimizer.step()
            losses.update(losses_D)
    finally:
        for p_, req_ in locals().get('d_req', []):
            p_.requires_grad_(req_)
        if trainer is None and saved_hook is not None:
            sgg_model._grad_ready_hook = saved_hook
    return {k: v.detach() for k, v in losses.items()}


class _LossCfg(object):
    """what Trainer.losses reads from `self` (loss form and weights; one process)"""

    def __init__(self, loss_type, loss_weights, world=1):
        self.loss_type, self.loss_weights, self.dist_on, self.world = loss_type, tuple(float(x) for x in loss_weights), world > 1, world
