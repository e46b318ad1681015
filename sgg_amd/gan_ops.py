"""GAN generator data movement: `boxes_to_layout` (augment/layout.py:33-71) and the gather / pooling steps of
`GraphTripleConv` (augment/graphconv.py:51-119) as HIP kernels behind the reference's signatures, differentiable; the graph
convolution's MLPs on sgg_amd/dense.py.  The model around them: sgg_amd/feature_gan.py.
"""
import torch

from . import _lib, ops
from .ops import _p, _stream, dt


class _BoxesToLayout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vecs, boxes, obj_to_img, H, W, avg, N):
        """vecs channels-last: [O,S,S,D] or [O,D]"""
        O, D = vecs.shape[0], vecs.shape[-1]
        S = vecs.shape[1] if vecs.dim() == 4 else 0
        obj_img = obj_to_img.to(torch.int32).contiguous()
        counts = torch.bincount(obj_to_img.long(), minlength=N).to(torch.int32) if avg else None
        boxes = boxes.float().contiguous()
        vecs = vecs.contiguous()
        out = torch.empty((N, H, W, D), dtype=vecs.dtype, device=vecs.device)
        _lib.call('sgg_boxes_to_layout_fwd', _p(vecs), _p(boxes, torch.float32), _p(obj_img), N, O, S, D, H, W, int(avg), _p(out),
                  dt(vecs), _stream())
        ctx.save_for_backward(boxes, obj_img, counts)
        ctx.geom = (O, S, D, H, W, int(avg), tuple(vecs.shape))
        return out

    @staticmethod
    def backward(ctx, d_out):
        boxes, obj_img, counts = ctx.saved_tensors
        O, S, D, H, W, avg, shape = ctx.geom
        d_out = d_out.contiguous()
        d_vecs = torch.empty(shape, dtype=d_out.dtype, device=d_out.device)
        _lib.call('sgg_boxes_to_layout_bwd', _p(d_out), _p(boxes), _p(obj_img), _p(counts), O, S, D, H, W, avg, _p(d_vecs), dt(d_out),
                  _stream())
        return d_vecs, None, None, None, None, None, None


def boxes_to_layout_nhwc(vecs, boxes, obj_to_img, H, W=None, pooling='sum', num_images=None):
    """channels-last form: vecs [O,S,S,D] or [O,D] -> [N,H,W,D] (what a following channels-last convolution consumes)."""
    if pooling not in ('sum', 'avg'):
        raise ValueError('Invalid pooling "%s"' % pooling)                                           # layout.py:163-164
    N = int(obj_to_img.max().item()) + 1 if num_images is None else int(num_images)                  # :150 (the reference syncs here too)
    return _BoxesToLayout.apply(vecs, boxes, obj_to_img, H, H if W is None else W, pooling == 'avg', N)


def boxes_to_layout(vecs, boxes, obj_to_img, H, W=None, pooling='sum'):
    """augment/layout.py:33-71, same arguments and result: vecs (O, D) or (O, D, S, S), boxes (O, 4) in [0, 1] as x0, y0, x1, y1,
    obj_to_img (O,) -> (N, D, H, W)."""
    if vecs.dim() == 4 and len(vecs.squeeze().shape) > 2:                                            # :56-60
        v = vecs.permute(0, 2, 3, 1)
    else:
        v = vecs.reshape(vecs.shape[0], vecs.shape[1])
    return boxes_to_layout_nhwc(v, boxes, obj_to_img, H, W, pooling).permute(0, 3, 1, 2)


class _TripleGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, obj_vecs, pred_vecs, edges):
        T, Din, De = edges.shape[0], obj_vecs.shape[1], pred_vecs.shape[1]
        out = torch.empty((T, 2 * Din + De), dtype=obj_vecs.dtype, device=obj_vecs.device)
        edges = edges.long().contiguous()
        _lib.call('sgg_triple_gather', _p(obj_vecs.contiguous()), _p(pred_vecs.contiguous()), _p(edges, torch.int64), T, Din, De, _p(out),
                  dt(out), _stream())
        ctx.save_for_backward(edges)
        ctx.geom = (obj_vecs.shape[0], Din, De)
        return out

    @staticmethod
    def backward(ctx, d):
        (edges,) = ctx.saved_tensors
        O, Din, De = ctx.geom
        # the adjoint of the gather is the (sum-)pooling of the two outer column blocks back onto the objects
        csr = ops.edge_csr(torch.cat((edges.new_zeros((edges.shape[0], 1)), edges), 1).contiguous(), O)
        d = d.contiguous()
        d_obj = torch.empty((O, Din), dtype=d.dtype, device=d.device)
        _lib.call('sgg_triple_pool_fwd', _p(d), d.shape[1], Din + De, _p(csr[0]), _p(csr[1]), _p(csr[2]), _p(csr[3]), O, Din, 0,
                  _p(d_obj), dt(d), _stream())
        return d_obj, d[:, Din:Din + De].contiguous(), None


class _TriplePool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rows, edges, O, Hd, o_off, avg):
        edges = edges.long().contiguous()
        csr = ops.edge_csr(torch.cat((edges.new_zeros((edges.shape[0], 1)), edges), 1).contiguous(), O)
        rows = rows.contiguous()
        pooled = torch.empty((O, Hd), dtype=rows.dtype, device=rows.device)
        _lib.call('sgg_triple_pool_fwd', _p(rows), rows.shape[1], o_off, _p(csr[0]), _p(csr[1]), _p(csr[2]), _p(csr[3]), O, Hd, int(avg),
                  _p(pooled), dt(rows), _stream())
        ctx.save_for_backward(edges, csr[0], csr[2])
        ctx.geom = (tuple(rows.shape), Hd, o_off, int(avg))
        return pooled

    @staticmethod
    def backward(ctx, d_pooled):
        edges, out_ptr, in_ptr = ctx.saved_tensors
        shape, Hd, o_off, avg = ctx.geom
        d_pooled = d_pooled.contiguous()
        d_rows = torch.zeros(shape, dtype=d_pooled.dtype, device=d_pooled.device)
        _lib.call('sgg_triple_pool_bwd', _p(d_pooled), _p(edges, torch.int64), _p(out_ptr), _p(in_ptr), shape[0], Hd, avg, shape[1],
                  o_off, _p(d_rows), dt(d_pooled), _stream())
        return d_rows, None, None, None, None, None


def triple_gather(obj_vecs, pred_vecs, edges):
    """graphconv.py:68-78: [obj[s] | pred | obj[o]] rows, (T, 2*Din + De)."""
    return _TripleGather.apply(obj_vecs, pred_vecs, edges)


def triple_pool(new_t_vecs, edges, num_objs, hidden_dim, o_off, pooling='avg'):
    """graphconv.py:93-115 on the net1 output itself: subject part = columns [0, H), object part = columns [o_off, o_off + H)."""
    assert pooling in ('sum', 'avg'), 'Invalid pooling "%s"' % pooling                               # :35
    return _TriplePool.apply(new_t_vecs, edges, num_objs, hidden_dim, o_off, pooling == 'avg')


def build_mlp(dim_list, activation='relu', batch_norm='none', dropout=0, final_nonlinearity=True):
    """graphconv.py:157-176: Linear [BatchNorm1d] [ReLU | LeakyReLU] ... with the reference's Sequential indices (checkpoints load).
    The Linear layers multiply on this package's GEMM (sgg_amd/dense.py), the normalisation is its row-matrix BatchNorm."""
    from . import dense
    act = {'relu': torch.nn.ReLU, 'leakyrelu': torch.nn.LeakyReLU}.get(activation)
    layers = []
    n_layers = len(dim_list) - 1
    for i, (d_in, d_out) in enumerate(zip(dim_list[:-1], dim_list[1:])):
        layers.append(dense.Linear(d_in, d_out))
        if i < n_layers - 1 or final_nonlinearity:
            if batch_norm == 'batch':
                layers.append(dense.BatchNormRows(d_out))
            if act is not None:
                layers.append(act())
        if dropout > 0:
            layers.append(torch.nn.Dropout(p=dropout))
    return torch.nn.Sequential(*layers)


def _init_weights(module):
    if isinstance(module, torch.nn.Linear):
        torch.nn.init.kaiming_normal_(module.weight)                                                 # graphconv.py:11-14


class GraphTripleConv(torch.nn.Module):
    """augment/graphconv.py:17-119, one scene-graph convolution: same constructor, parameter names and outputs."""

    def __init__(self, input_dim, input_edge_dim=None, output_dim=None, hidden_dim=512, pooling='avg', mlp_normalization='none',
                 final_nonlinearity=True):
        super(GraphTripleConv, self).__init__()
        output_dim = input_dim if output_dim is None else output_dim
        input_edge_dim = input_dim if input_edge_dim is None else input_edge_dim
        self.input_dim, self.output_dim, self.hidden_dim = input_dim, output_dim, hidden_dim
        self.final_nonlinearity = final_nonlinearity
        assert pooling in ['sum', 'avg'], 'Invalid pooling "%s"' % pooling
        self.pooling = pooling
        self.net1 = build_mlp([2 * input_dim + input_edge_dim, hidden_dim, 2 * hidden_dim + output_dim],
                              batch_norm=mlp_normalization, final_nonlinearity=final_nonlinearity)
        self.net1.apply(_init_weights)
        self.net2 = build_mlp([hidden_dim, hidden_dim, output_dim], batch_norm=mlp_normalization,
                              final_nonlinearity=final_nonlinearity)
        self.net2.apply(_init_weights)

    def forward(self, obj_vecs, pred_vecs, edges):
        """obj_vecs (O, Din), pred_vecs (T, De), edges (T, 2) -> new_obj_vecs (O, Dout), new_pred_vecs (T, Dout)"""
        H, Dout = self.hidden_dim, self.output_dim
        new_t = self.net1(triple_gather(obj_vecs, pred_vecs, edges))                                 # :68-79
        if not self.final_nonlinearity:                                                              # :86-88
            new_t = torch.cat((torch.relu(new_t[:, :H]), new_t[:, H:H + Dout], torch.relu(new_t[:, H + Dout:])), 1)
        pooled = triple_pool(new_t, edges, obj_vecs.shape[0], H, H + Dout, self.pooling)             # :93-115
        return self.net2(pooled), new_t[:, H:H + Dout]


class GraphTripleConvNet(torch.nn.Module):
    """augment/graphconv.py:120-153: a stack of scene-graph convolutions (`G_gcn` of augment/gan.py:109-115)."""

    def __init__(self, input_dim, input_edge_dim=None, output_dim=None, num_layers=5, hidden_dim=512, pooling='avg',
                 mlp_normalization='none'):
        super(GraphTripleConvNet, self).__init__()
        self.num_layers = num_layers
        self.gconvs = torch.nn.ModuleList()
        for i in range(num_layers):
            self.gconvs.append(GraphTripleConv(input_dim if i == 0 else hidden_dim,
                                               input_edge_dim=input_edge_dim if i == 0 else hidden_dim,
                                               output_dim=output_dim if i == num_layers - 1 else hidden_dim,
                                               hidden_dim=hidden_dim, pooling=pooling, mlp_normalization=mlp_normalization,
                                               final_nonlinearity=i < num_layers - 1))

    def forward(self, obj_vecs, pred_vecs, edges):
        assert len(edges.shape) == 2 and edges.shape[1] == 2, edges.shape
        for gconv in self.gconvs:
            obj_vecs, pred_vecs = gconv(obj_vecs, pred_vecs, edges)
        return obj_vecs, pred_vecs
