"""Training-mode forward and backward of the trainable relation head on the HIP path.

Covers everything `RelModelStanford.predict` does in `model.train()` (sgg_models/rel_model_stanford.py:97-107 with the
classifier Dropouts of rel_model_base.py:110-111 and the batch-statistic BatchNorms of lib/get_union_boxes.py:54,58)
and its gradient w.r.t. every trainable parameter.  RoI features are leaves (the detector is frozen and `fmap` is
detached, rel_model_stanford.py:131), so no gradient flows into RoIAlign.  The whole head is ONE autograd node
(`PredictFn`): the reference's losses (lib/losses.py) and optimiser (lib/pytorch_misc.py:130-157) run on top of it
unchanged.

Dense gradient contractions reuse the MFMA GEMM:  dX = dY . W  is  gemm(dY, W^T)  with W^T prepared per step,
dW = dY^T . X  is  gemm(dY^T, X^T)  on transposed copies.  GRU weights are shared by the 4 calls of each cell; their
gradients are single GEMMs over the row-stacked calls (no accumulation passes).
"""
import os

import torch

from . import _lib, ops
from .imp import GATES, node_lane

DROPOUT_P = 0.5  # nn.Dropout() default in torchvision's VGG classifier


def param_names(model):
    fc, _ = model.fc_layers()        # the four big Linear layers under their own names (VGG classifier / TwoMLPHead copies)
    names = [fc[k][0] + sfx for k in ('fc6_edge', 'fc7_edge', 'fc6_obj', 'fc7_obj') for sfx in ('.weight', '.bias')]
    names += ['union_boxes.conv.0.weight', 'union_boxes.conv.0.bias', 'union_boxes.conv.2.weight',
             'union_boxes.conv.2.bias', 'union_boxes.conv.4.weight', 'union_boxes.conv.4.bias',
             'union_boxes.conv.6.weight', 'union_boxes.conv.6.bias', 'rel_fc.weight', 'rel_fc.bias', 'obj_fc.weight',
             'obj_fc.bias', 'obj_unary.weight', 'obj_unary.bias', 'edge_unary.weight', 'edge_unary.bias']
    for g in ('edge_gru', 'node_gru'):
        names += [g + '.weight_ih', g + '.weight_hh', g + '.bias_ih', g + '.bias_hh']
    for g in GATES:
        names += [g + '.0.weight', g + '.0.bias']
    return names


def _pad_cols(x, mult, dtype):
    M, N = x.shape
    Np = (N + mult - 1) // mult * mult
    if Np == N and x.dtype == dtype and x.is_contiguous():
        return x
    buf = torch.zeros((M, Np), dtype=dtype, device=x.device)
    buf[:, :N].copy_(x)     # cast + pad copy (data movement only)
    return buf


USE_TN = True      # tests flip this to cross-check the TN kernel against the transposes + NT path at benchmark grid sizes


def tn_gemm(X, Y, out=None, want_colsum=False, out_dtype=torch.float32):
    """X[M,a]^T . Y[M,b] -> f32 [a,b]  (weight gradients).  Rows are zero-padded to a multiple of 64 by the transposes.
    want_colsum: also return colsum(X) (the bias gradient when X = dY), computed inside X's transpose pass."""
    if (out is None and USE_TN and ops.split3_on() and not ops._BWD16[0] and ops.gemm_tn_x3_ok(X, Y)):
        # x3 mode: the same contraction on PAIR operands, both read as they lie (round 6: no fp32 transposes, no split of transposed copies)
        dw = ops.gemm_tn_x3(X, Y, out_dtype=out_dtype)
        return (dw, ops.colsum(X)) if want_colsum else dw
    if out is None and USE_TN and ops.gemm_tn_ok(X, Y):
        # 16-bit, aligned shapes: the TN kernel reads dY and X as they lie (no transposed copies); bias gradient = colsum(dY)
        dw = ops.gemm_tn(X, Y, out_dtype=out_dtype)
        return (dw, ops.colsum(X)) if want_colsum else dw
    if want_colsum:
        xt, cs = ops.transpose(X, want_colsum=True)
        return ops.gemm(xt, ops.transpose(Y), out_dtype=out_dtype, out=out), cs
    return ops.gemm(ops.transpose(X), ops.transpose(Y), out_dtype=out_dtype, out=out)


TRANSPOSED = ('fc7_obj', 'obj_unary', 'fc7_edge', 'edge_unary', 'obj_fc', 'rel_fc')      # W^T copies for the dX GEMMs of the backward


def _transposed_operands(model, w, keep, t):
    """the backward-only operands: W^T copies (dX = dY W runs as an NT GEMM on W^T), written in place into `keep`'s buffers"""
    batch = []          # 16-bit modes: every W^T copy whose buffer exists already is rewritten by ONE launch (ops.transpose_multi) at the end

    def tr(name, x):
        out = keep.get(name)
        if out is not None and (out.shape[0] != x.shape[1] or out.shape[1] < x.shape[0]):
            out = None
        if out is not None and ops.is_half(x) and out.dtype == x.dtype and x.stride(1) == 1 and MULTI_TRANSPOSE:
            batch.append((x, out))
            return out
        keep[name] = ops.transpose(x, out=out)
        return keep[name]
    for name in TRANSPOSED:
        t[name + '_t'] = tr(name + '_t', w[name])                        # [K, Np]
    t['w6sum_t'] = tr('w6sum_t', w['fc6_edge_sum'])                      # [C, 4096]
    imp = w['imp']
    for g in ('edge_gru', 'node_gru'):
        t[g + '_w_ih_t'] = tr(g + '_w_ih_t', getattr(imp, g + '_w_ih'))  # [H, 3H]
        t[g + '_w_hh_t'] = tr(g + '_w_hh_t', getattr(imp, g + '_w_hh'))
    t['rc_w2_t'] = tr('rc_w2_t', t['rc_w2'])                              # [d2, d]
    if batch:
        ops.transpose_multi(batch)


MULTI_TRANSPOSE = os.environ.get('SGG_MULTI_TRANSPOSE', '1') != '0'


def rebuild_transposes(model):
    """The W^T copies again from the current operands, in place (no allocation once they exist).  A replayed step (sgg_amd/graph_step.py)
    makes them on the lane beside the edge branch's forward MLP instead of inside the update: they are read by the backward only, and the
    update is the long pole of the update || VGG-forward pair."""
    w = model.prepared()
    _transposed_operands(model, w, model.__dict__['_train_bufs'], w['train'])


def train_weights(model, transposes=True):
    """Forward operands (model.prepared()) + the transposed copies the backward needs, cached on the same key.  The derived buffers
    are allocated once per (model, dtype) and rewritten in place after every update: a rebuild is the library's transposes plus two
    strided copies, no allocation and no fill.  transposes=False (a replayed step's update graph): the W^T copies keep their buffers but are
    NOT rewritten here -- the caller has rebuild_transposes() run before the backward reads them."""
    w = model.prepared()
    if 'train' in w:
        return w
    dt = model.compute_dtype
    keep = model.__dict__.setdefault('_train_bufs', {})
    if keep.get('dtype') != dt or keep.get('device') != model.rel_fc.weight.device:
        keep.clear()
        keep.update(dtype=dt, device=model.rel_fc.weight.device)
    t = {}
    ub = model.union_boxes
    f = lambda p: p.detach().float().contiguous()
    c0, c4 = ub.conv[0].weight, ub.conv[4].weight
    d2, d = c0.shape[0], c4.shape[0]
    if d2 % 64:
        raise NotImplementedError('training needs union_boxes dim/2 to be a multiple of 64')
    w1, w2 = keep.get('rc_w1'), keep.get('rc_w2')
    if w1 is None or tuple(w1.shape) != (d2, 128) or tuple(w2.shape) != (d, d2):
        w1 = keep['rc_w1'] = torch.zeros((d2, 128), dtype=dt, device=c0.device)     # 7x7x2 taps = 98 columns of the 128-wide patch rows
        w2 = keep['rc_w2'] = torch.empty((d, d2), dtype=dt, device=c0.device)
    with torch.no_grad():
        w1[:, :98].copy_(c0.detach().reshape(d2, 98))                       # (one converting strided copy each)
        w2.copy_(c4.detach()[:, :, 1, 1])                                   # centre tap (the only one that sees data)
    t['rc_w1'], t['rc_w2'] = w1, w2
    t['rc_b1'] = f(ub.conv[0].bias)
    t['rc_b2'] = f(ub.conv[4].bias)
    names = [n + '_t' for n in TRANSPOSED] + ['w6sum_t', 'rc_w2_t'] + [g + s_ for g in ('edge_gru', 'node_gru') for s_ in ('_w_ih_t', '_w_hh_t')]
    if transposes or any(keep.get(n) is None for n in names):
        _transposed_operands(model, w, keep, t)
    else:
        for n in names:
            t[n] = keep[n]                                                  # (stale until rebuild_transposes)
    t['rc_g1'], t['rc_be1'] = f(ub.conv[2].weight), f(ub.conv[2].bias)
    t['rc_g2'], t['rc_be2'] = f(ub.conv[6].weight), f(ub.conv[6].bias)
    t['ones'] = None
    w['train'] = t
    return w


def _imp_lane(dev):
    """the node lane's stream for the node cell of the training loop (SGG_TRAIN_IMP_LANE=0: everything on one stream)"""
    if os.environ.get('SGG_TRAIN_IMP_LANE', '1') == '0':
        return None
    lane = node_lane(dev)
    return lane[0] if lane is not None else None


def _gru_fwd(x, h, w_ih, w_hh, b_ih, b_hh, dtype, out, dot_w=None):
    """-> (gi, gh, dots): dots f32[M,4] = the new state's gate dot products against dot_w (None without dot_w)."""
    gi = ops.gemm(x, w_ih, b_ih, out_dtype=torch.float32)
    gh = ops.gemm(h, w_hh, b_hh, out_dtype=torch.float32) if h is not None else None
    r = ops.gru_gate(gi, gh, b_hh if h is None else None, h, dtype, out=out, dot_w=dot_w)
    return gi, gh, (r[1] if dot_w is not None else None)


class PredictFn(torch.autograd.Function):
    """(obj_dists, rel_dists) = predict(node_feat, edge_feat, ...) in training mode; gradients for all head parameters."""

    @staticmethod
    def forward(ctx, model, nf, ef, rois, rel_inds, im_inds, seed, dropout_p, *params):
        ev = getattr(model, '_operands_ready', None)     # trainer: operands rebuilt on a side stream under the VGG forward
        if ev is not None:
            torch.cuda.current_stream(nf.device).wait_event(ev)
            model._operands_ready = None
        w = train_weights(model)
        t, imp, dt = w['train'], w['imp'], model.compute_dtype
        N, E, H = nf.shape[0], rel_inds.shape[0], model.hidden_dim
        paired = getattr(model, '_pairing_hint', None)   # ef holds the rows of the unordered box pairs (sgg_amd/pairing.py)
        model._pairing_hint = None
        assert ef.shape[0] == (paired.U if paired is not None else E), (ef.shape, E)
        ub = model.union_boxes
        dev = nf.device
        sv = {}
        # ---- rect conv, batch-statistics BatchNorm (lib/get_union_boxes.py:51-59)
        pairs = ops.pairs_of(rel_inds)
        patches = ops.union_rect_patches(rois, pairs, dt, ub.pooling_size * 4 - 1, 128,
                                         im_sizes=ub.raster_sizes(getattr(model, '_im_sizes_hint', None)))   # [4E,128]
        h1 = ops.gemm(patches, t['rc_w1'], t['rc_b1'], ops.ACT_RELU)                               # [4E,d2]
        bn1, bn2 = ub.conv[2], ub.conv[6]
        bn_sync = getattr(model, '_bn_sync', None)      # DP trainer: batch statistics over the rows of every rank
        h2, arg, m1, is1 = ops.bn_train(h1, t['rc_g1'], t['rc_be1'], bn1.running_mean, bn1.running_var, bn1.eps,
                                        bn1.momentum, True, bn_sync)                               # [E,d2]
        h3 = ops.gemm(h2, t['rc_w2'], t['rc_b2'], ops.ACT_RELU)                                    # [E,d]
        rect, _, m2, is2 = ops.bn_train(h3, t['rc_g2'], t['rc_be2'], bn2.running_mean, bn2.running_var, bn2.eps,
                                        bn2.momentum, False, bn_sync)
        ub.count_train_batch()                    # BatchNorm's num_batches_tracked, applied when somebody looks (union_boxes.py)
        sv.update(patches=patches, h1=h1, h2=h2, arg=arg, m1=m1, is1=is1, h3=h3, m2=m2, is2=is2, rect=rect)
        # (edge_feat + conv(rects))^T, the X operand of fc6's weight gradient, depends on forward data only: its 0.25 ms of
        # HBM-bound transposing runs NOW on the node lane's stream, under the MFMA-bound fc6 GEMMs below, instead of on the
        # backward's critical path.  The buffer is allocated under that stream and handed back to it when sv is dropped.
        from .imp import node_lane
        lane = node_lane(dev)
        # unordered pairs, 16-bit: fc6's weight gradient runs on the TN form of the ping-pong kernel, straight from the pooled rows
        sv['tn6'] = paired is not None and ef.dim() == 2 and w['fc6_edge'].shape[0] % 256 == 0 and ops.gemm_tn256_ok(ef, ef)
        if lane is not None and ops.is_half(dt):
            side, ev_main, _ = lane
            ev_main.record(torch.cuda.current_stream(dev))
            side.wait_event(ev_main)                                 # ef and rect are ready
            with torch.cuda.stream(side):
                PPs = model.pool_sz ** 2
                if sv.get('tn6'):
                    pass                    # the weight gradient reads the pooled rows as they lie (sgg_gemm_tn256): no transposed copy
                elif paired is not None:    # the rect term does not ride here: it differs between a pair's two edges (backward)
                    sv['x6t'] = ops.transpose(ef)
                else:
                    sv['x6t'] = ops.transpose(ef, add=rect.float() if rect.dtype != torch.float32 else rect, group=PPs)
                ev = torch.cuda.Event()
                ev.record(side)
            sv['x6t_ready'] = ev
        # row-stacked inputs / states of the 4 calls of each GRU cell
        T = model.mp_iter
        XN = torch.empty(((T + 1) * N, H), dtype=dt, device=dev)      # inputs of the node cell: obj_rep, ctx_0 .. ctx_{T-1}
        # XH = [rel_rep (E rows) ; v_0 .. v_T (N rows each)]: the edge cell's input weight W_ih sees rel_rep in its first call and the node
        # states v_i in the others (the node projection, csrc/imp.hip) -- one contiguous operand for its weight-gradient contraction
        XH = torch.empty((E + (T + 1) * N, H), dtype=dt, device=dev)
        XE0, HN = XH[:E], XH[E:]
        # block c = vert_c / edge_c: the hidden state LEAVING call c = entering call c+1.  (The state entering call 0 is zero: that
        # call adds nothing to the hidden-weight gradients, whose contractions therefore run over the rows of calls 1..T only.)
        HE = torch.empty(((T + 1) * E, H), dtype=dt, device=dev)
        # a capture in progress (sgg_amd/graph_step.py): the node branch's MLP (256 rows against 25088-wide weights: HBM- and latency-bound) becomes
        # a graph of its own, replayed on the lane's stream beside the edge branch's MFMA-bound MLP; the two meet in front of the message passing
        fsplit = getattr(model, '_graph_split', None)
        if fsplit is not None:
            fsplit.next('lane')
        # ---- nodes: obj_unary(roi_fmap_obj(node_feat))  (Linear ReLU Dropout Linear ReLU Dropout)
        _lib.set_tag('fc6_obj')
        x6 = ops.gemm(nf, w['fc6_obj'], w['fc6_obj_b'], ops.ACT_RELU)
        _lib.set_tag('mlp')
        if dropout_p > 0:
            ops.dropout_(x6, dropout_p, seed, 1)
        x7 = ops.gemm(x6, w['fc7_obj'], w['fc7_obj_b'], ops.ACT_RELU)
        if dropout_p > 0:
            ops.dropout_(x7, dropout_p, seed, 2)
        ops.gemm(x7, w['obj_unary'], w['obj_unary_b'], out=XN[:N])
        if fsplit is not None:
            if getattr(model, '_transposes_in_forward', False):
                rebuild_transposes(model)        # (the replayed step's update graph left them out: made here, on the lane, read by the backward)
            fsplit.next('main')
        # ---- edges: relu(edge_unary(roi_fmap(edge_feat + conv(rects))))  (Linear ReLU Dropout Linear)
        _lib.set_tag('fc6_edge')
        if paired is not None:       # the long contraction once per unordered pair (f32), then per edge: + rect term + bias, ReLU
            yu = ops.gemm(ef, w['fc6_edge'], out_dtype=torch.float32)
            _lib.set_tag('fc6_edge_rect')
            y6 = ops.gemm_addrows(rect, w['fc6_edge_sum'], w['fc6_edge_b'], yu, paired.e2u, ops.ACT_RELU)
            del yu
        else:
            y6 = ops.gemm(ef, w['fc6_edge'], w['fc6_edge_b'], ops.ACT_RELU, A2=rect, W2=w['fc6_edge_sum'])
        _lib.set_tag('mlp')
        if dropout_p > 0:
            ops.dropout_(y6, dropout_p, seed, 3)
        edge_relu7 = model.fc_layers()[1]          # TwoMLPHead copies (resnet50): ReLU after the edge branch's fc7 as well
        y7 = ops.gemm(y6, w['fc7_edge'], w['fc7_edge_b'], ops.ACT_RELU if edge_relu7 else ops.ACT_NONE)
        ops.gemm(y7, w['edge_unary'], w['edge_unary_b'], ops.ACT_RELU, out=XE0)
        sv.update(x6=x6, x7=x7, y6=y6, y7=y7)
        if fsplit is not None:
            fsplit.next('joined')
        # ---- message passing (rel_model_stanford.py:68-94) with the node projection: per iteration ctx = read stream over e_i,
        # P_i = v_i W_ih^T, e_{i+1} = gate kernel on (e_i W_hh^T, P_i[s], P_i[o]); the gate dot products come out of the gate kernels
        _lib.set_tag('imp')
        csr = getattr(model, '_csr_hint', None)       # index tables cached per box-count signature (rel_model_stanford.forward)
        model._csr_hint = None
        if csr is None:
            csr = ops.edge_csr(rel_inds, N, im_inds, graphs=getattr(model, '_graphs_hint', None))
        if T > 0 and not ops.gate_dots_ok(H):
            raise NotImplementedError('message passing needs hidden_dim / 8 to be a power of two <= 64 (got hidden_dim %d)' % H)
        wv, we = (imp.gate_w[:, :H], imp.gate_w[:, H:]) if T > 0 else (None, None)
        gin, ghn, ghe, Ps, nds, eds = [], [], [], [], [], []
        # The node cell is three launches on 32 B rows -- pure launch / DRAM latency -- and depends on (ctx_i, v_i) only: on the node lane's
        # stream it runs beside the edge cell of the same iteration (as in message_pass of the evaluation path) instead of after it.
        main = torch.cuda.current_stream(dev)
        side = _imp_lane(dev)
        node_done = None

        def node_cell(x, h, out, dots_w, after):
            """-> (gi, gh, dots) of the node GRU; on the lane when there is one (`after`: main-stream event its inputs wait for)"""
            nonlocal node_done
            if side is None:
                return _gru_fwd(x, h, imp.node_gru_w_ih, imp.node_gru_w_hh, imp.node_gru_b_ih, imp.node_gru_b_hh, dt, out, dots_w)
            side.wait_event(after)
            with torch.cuda.stream(side):
                res = _gru_fwd(x, h, imp.node_gru_w_ih, imp.node_gru_w_hh, imp.node_gru_b_ih, imp.node_gru_b_hh, dt, out, dots_w)
                node_done = torch.cuda.Event()
                node_done.record(side)
            for r_ in res:
                if r_ is not None:
                    r_.record_stream(main)            # allocated under the lane's stream, read (and freed) under the main one
            return res

        def mark(stream=None):
            ev_ = torch.cuda.Event()
            ev_.record(stream or main)
            return ev_

        a, b, nd = node_cell(XN[:N], None, HN[:N], wv, mark() if side is not None else None)
        gin.append(a); ghn.append(b)
        gie0, _, ed = _gru_fwd(XE0, None, imp.edge_gru_w_ih, imp.edge_gru_w_hh, imp.edge_gru_b_ih, imp.edge_gru_b_hh, dt, HE[:E], we)
        for i in range(T):
            v_i, e_i = HN[i * N:(i + 1) * N], HE[i * E:(i + 1) * E]
            more = i + 1 < T
            ctx_i = XN[(i + 1) * N:(i + 2) * N]                                # ctx = ctx_out + ctx_in (kept for d W_ih of the node cell)
            if node_done is not None:
                main.wait_event(node_done)                                    # v_i and its gate dot products
            ops.imp_ctx(e_i, csr, N, nd, ed, imp.gate_b, pair=2, ctx_sum=ctx_i)
            a, b, nd_new = node_cell(ctx_i, v_i, HN[(i + 1) * N:(i + 2) * N], wv if more else None, mark() if side is not None else None)
            P = ops.gemm(v_i, imp.edge_gru_w_ih, None, out_dtype=torch.float32)
            gh = ops.gemm(e_i, imp.edge_gru_w_hh, imp.edge_gru_b_hh, out_dtype=ops.gh_dtype(dt))       # (saved for the backward in that type too)
            r = ops.gru_gate_proj(gh, P, imp.edge_gru_b_ih, csr, nd, ed, imp.gate_b, e_i, out=HE[(i + 1) * E:(i + 2) * E],
                                  dot_w=we if more else None)
            ghe.append(gh); Ps.append(P); nds.append(nd); eds.append(ed)
            gin.append(a); ghn.append(b)
            nd, ed = nd_new, (r[1] if more else None)
        if node_done is not None:
            main.wait_event(node_done)
        vT, eT = HN[T * N:(T + 1) * N], HE[T * E:(T + 1) * E]
        _lib.set_tag('heads')
        obj = ops.gemm(vT, w['obj_fc'], w['obj_fc_b'], out_dtype=torch.float32)
        rel = ops.gemm(eT, w['rel_fc'], w['rel_fc_b'], out_dtype=torch.float32)
        sv.update(paired=paired, XN=XN, XH=XH, HN=HN, HE=HE, gin=gin, ghn=ghn, gie0=gie0, ghe=ghe, Ps=Ps, nds=nds, eds=eds, csr=csr, nf=nf,
                  ef=ef, rel_inds=rel_inds, N=N, E=E, H=H, dropout_p=dropout_p)
        _lib.set_tag('')
        ctx.model, ctx.sv = model, sv
        return obj, rel

    @staticmethod
    def backward(ctx, d_obj, d_rel):
        # x3 mode with an f16 backward (ops.set_backward_f16): the contractions below round their operands to f16 once instead of splitting them
        prev = ops.set_backward_f16(bool(getattr(ctx.model, 'backward_f16', False)) and ops.split3_on())
        try:
            return PredictFn._backward(ctx, d_obj, d_rel)
        finally:
            ops.set_backward_f16(prev)

    @staticmethod
    def _backward(ctx, d_obj, d_rel):
        model, sv = ctx.model, ctx.sv
        w = train_weights(model)
        t, imp, dt = w['train'], w['imp'], model.compute_dtype
        if dt == torch.float16 and not getattr(model, '_loss_scaled', False) and not getattr(model, '_warned_f16_backward', False):
            # f16 activation gradients have five exponent bits: the Trainer scales the loss (and un-scales in the optimiser); a plain
            # `loss.backward(); optimizer.step()` loop does not -- small gradients flush to zero without any error
            import warnings
            warnings.warn('sgg_amd: f16 backward outside sgg_amd.trainer.Trainer -- no loss scale is applied; scale the loss yourself and set '
                          'model._loss_scaled = True, or train with set_compute_dtype(torch.bfloat16)', RuntimeWarning)
            model._warned_f16_backward = True
        N, E, H, T = sv['N'], sv['E'], sv['H'], model.mp_iter
        XN, XH, HN, HE = sv['XN'], sv['XH'], sv['HN'], sv['HE']
        csr = sv['csr']
        dev = XN.device
        G = {}
        rows = lambda buf, c, n: buf[c * n:(c + 1) * n]
        # Data-parallel hooks (set by the trainer): a big gradient is handed over the moment it exists so that its
        # all-reduce overlaps the rest of the backward; with a bf16 wire the GEMM emits the wire dtype directly.
        grad_hook = getattr(model, '_grad_ready_hook', None)
        wire = getattr(model, '_grad_wire_dtype', None) if grad_hook is not None else None
        handed = set()

        def big_dtype():
            return wire if wire is not None else torch.float32

        def hook(name):
            if grad_hook is None:
                return
            if grad_hook(name, G[name]) and G[name].dtype != torch.float32:
                handed.add(name)                 # the all-reduce owns the wire-dtype buffer; autograd gets nothing
            elif G[name].dtype != torch.float32:
                G[name] = G[name].float()

        # Order: (A) the dX chain alone -- the critical path down to the two fc6 inputs; (B) the fc6 weight gradients
        # (83 % of all gradient bytes), each followed by its hook; (C) every other weight gradient, deferred to here so
        # that it runs while the fc6 all-reduces occupy the links.
        deferred = []
        _bl = node_lane(dev) if os.environ.get('SGG_BWD_LANE', '1') != '0' else None
        bwd_lane = _bl[0] if _bl is not None else None

        def lin_bwd(dY, X, Wt, name, tag, want_dx=True, big=False, n_out=None):
            """Y = X W^T + b : returns dX now; dW (f32 [N,K]) and db are computed in phase C.  n_out: dY arrives already
            zero-padded to a multiple of 128 columns (the fused loss kernel writes it so) and only its first n_out are real."""
            prepadded = n_out is not None
            # The E-row fc7 weight gradient ([4096 x 4096] over 7936 rows): the TN kernel (no transposed copies) takes 287 us, the ping-pong
            # kernel on ready transposes 223 us (tools/tn_bench.py) -- and the two transposes (HBM-bound, 65 MB each way) cost nothing when
            # the node lane's stream makes them NOW, while the main stream is still busy with the dX chain
            tposed = None
            # (without the lane the same contraction runs with the transposes in line: the lane is scheduling only, never a different sum)
            if (big and ops.is_half(dt) and not prepadded and dY.shape[0] >= 4096 and dY.shape[0] % 64 == 0 and os.environ.get('SGG_DW_NT', '1') != '0'
                    and not ops.gemm_tn256_ok(dY, X)):      # (whole 256 x 256 tiles: tn_gemm below takes the ping-pong kernel's TN form, no copies)
                if bwd_lane is not None:
                    ready_ = torch.cuda.Event()
                    ready_.record(torch.cuda.current_stream(dev))
                    bwd_lane.wait_event(ready_)
                    with torch.cuda.stream(bwd_lane):
                        tposed = (ops.transpose(dY), ops.transpose(X), torch.cuda.Event())
                        tposed[2].record(bwd_lane)
                else:
                    tposed = (ops.transpose(dY), ops.transpose(X), None)

            def dw():
                _lib.set_tag(tag)
                if tposed is not None:
                    if tposed[2] is not None:
                        main_ = torch.cuda.current_stream(dev)
                        main_.wait_event(tposed[2])
                        tposed[0].record_stream(main_)
                        tposed[1].record_stream(main_)
                    G[name + '.weight'] = ops.gemm(tposed[0], tposed[1], out_dtype=big_dtype())
                    G[name + '.bias'] = ops.colsum(dY)
                    hook(name + '.weight')
                    return
                n_out = dY.shape[1] if not prepadded else n_out_
                if (n_out % 128 or prepadded) and ops.is_half(dt) and not big:
                    # narrow heads (151 / 51 outputs): zero-pad dY to 128 columns so that the TN kernel takes it
                    dYp = dY if prepadded else _pad_cols(dY, 128, dt)
                    if USE_TN and ops.gemm_tn_ok(dYp, X):
                        G[name + '.weight'] = ops.gemm_tn(dYp, X)[:n_out].contiguous()
                        G[name + '.bias'] = ops.colsum(dYp)[:n_out].contiguous()
                        return
                dYr = dY[:, :n_out].contiguous() if prepadded else dY
                G[name + '.weight'], G[name + '.bias'] = tn_gemm(dYr, X, want_colsum=True,
                                                                 out_dtype=big_dtype() if big else torch.float32)
                if big:
                    hook(name + '.weight')
            n_out_ = n_out
            deferred.append((name, dw))
            if not want_dx:
                return None
            if prepadded:                                   # W^T is padded to a multiple of 64 columns: a strided view of dY matches it
                return ops.gemm(dY[:, :Wt.shape[1]], Wt)
            return ops.gemm(_pad_cols(dY, 64, dt), Wt)

        # ---- phase A
        _lib.set_tag('bwd_heads')
        pre = getattr(model, '_logit_grads', None)       # Trainer's fused loss: (d_obj [N,256], d_rel [E,128]) in dt, zero-padded
        model._logit_grads = None
        if pre is not None and ops.is_half(dt) and pre[0].dtype == dt and pre[0].shape[0] == N and pre[1].shape[0] == E:
            d_v = lin_bwd(pre[0], rows(HN, T, N), t['obj_fc_t'], 'obj_fc', 'bwd_heads', n_out=d_obj.shape[1])
            d_e = lin_bwd(pre[1], rows(HE, T, E), t['rel_fc_t'], 'rel_fc', 'bwd_heads', n_out=d_rel.shape[1])
        else:
            d_obj = d_obj.contiguous().to(dt) if d_obj.dtype != dt else d_obj.contiguous()
            d_rel = d_rel.contiguous().to(dt) if d_rel.dtype != dt else d_rel.contiguous()
            d_v = lin_bwd(d_obj, rows(HN, T, N), t['obj_fc_t'], 'obj_fc', 'bwd_heads')
            d_e = lin_bwd(d_rel, rows(HE, T, E), t['rel_fc_t'], 'rel_fc', 'bwd_heads')
        # IMP backward (rel_model_stanford.py:74-92 in reverse)
        _lib.set_tag('bwd_imp')
        dGIn = torch.empty(((T + 1) * N, 3 * H), dtype=dt, device=dev)
        dGHn = torch.empty(((T + 1) * N, 3 * H), dtype=dt, device=dev)
        # d_gi of the edge cell, laid out for its two parameter gradients without a copy: rows [0, T E) = calls T .. 1 (call c at
        # (T - c) E), rows [T E, (T+1) E) = call 0, then T N rows dP_0 .. dP_{T-1} (gradients of the node projections).
        # b_ih acts in every call of every edge: its gradient is the column sum of the first (T+1) E rows; W_ih saw rel_rep in call 0
        # and v_i in the projections: its gradient is the contraction of the LAST E + T N rows with XH = [rel_rep ; v_0 .. v_{T-1}].
        DG = torch.empty(((T + 1) * E + T * N, 3 * H), dtype=dt, device=dev)
        dgi_call = lambda c: DG[(T - c) * E:(T - c + 1) * E]
        dP = lambda i: DG[(T + 1) * E + i * N:(T + 1) * E + (i + 1) * N]
        dGHe = torch.empty(((T + 1) * E, 3 * H), dtype=dt, device=dev)
        d_gw = torch.empty((4, 2 * H), dtype=torch.float32, device=dev)      # both halves are written whole in phase C
        # gate-side partials of all iterations, stacked like HN / HE so that the gate-weight gradients are three
        # contractions over T E / T N rows in phase C instead of 3 T small ones
        da_all = torch.empty((T * E, 4), dtype=torch.float32, device=dev)
        nsum_all = torch.empty((T * N, 4), dtype=torch.float32, device=dev)
        main_s = torch.cuda.current_stream(dev)
        imp_side = _imp_lane(dev)
        for i in range(T - 1, -1, -1):
            v_i, e_i = rows(HN, i, N), rows(HE, i, E)
            nd_i, ed_i = sv['nds'][i], sv['eds'][i]
            # v_{i+1} = GRU_n(ctx_i, v_i): four launches on 32 B rows, independent of the edge cell's backward below -- on the node lane
            def node_part(d_v=d_v, v_i=v_i, i=i):
                d_v_prev = ops.gru_gate_bwd(d_v, sv['gin'][i + 1], sv['ghn'][i + 1], None, v_i, rows(dGIn, i + 1, N),
                                            rows(dGHn, i + 1, N))
                d_ctx = ops.gemm(rows(dGIn, i + 1, N), t['node_gru_w_ih_t'])
                ops.add_(d_v_prev, ops.gemm(rows(dGHn, i + 1, N), t['node_gru_w_hh_t']))
                return d_v_prev, d_ctx
            if imp_side is None:
                d_v_prev, d_ctx = node_part()
            else:
                ev_in = torch.cuda.Event()
                ev_in.record(main_s)
                imp_side.wait_event(ev_in)                      # d_v (the previous iteration's join)
                with torch.cuda.stream(imp_side):
                    d_v_prev, d_ctx = node_part()
                    ev_node = torch.cuda.Event()
                    ev_node.record(imp_side)
                d_v_prev.record_stream(main_s)
                d_ctx.record_stream(main_s)
            # e_{i+1} = GRU_e(g_sub P_i[s] + g_obj P_i[o] + b_ih ; e_i)
            d_e_prev, dq = ops.gru_gate_proj_bwd(d_e, sv['ghe'][i], sv['Ps'][i], imp.edge_gru_b_ih, csr, nd_i, ed_i, imp.gate_b, e_i,
                                                 dgi_call(i + 1), rows(dGHe, i + 1, E))
            ops.add_(d_e_prev, ops.gemm(rows(dGHe, i + 1, E), t['edge_gru_w_hh_t']))
            if imp_side is not None:
                main_s.wait_event(ev_node)
            # the four gates and the context sums
            da = ops.imp_edge_ctx_bwd(e_i, csr, nd_i, ed_i, imp.gate_w, imp.gate_b, dq, d_ctx, d_e_prev, da=rows(da_all, i, E))
            ops.imp_node_gates_bwd(da, csr, imp.gate_w, d_v_prev, nsum=rows(nsum_all, i, N))
            # the node projection: dP_i[n] = sum_{s(e)=n} g_sub d_gi[e] + sum_{o(e)=n} g_obj d_gi[e]  (the forward's read stream on
            # the d_gi rows with the other gate pair), then through W_ih into d_v
            ops.imp_ctx(dgi_call(i + 1), csr, N, nd_i, ed_i, imp.gate_b, pair=0, ctx_sum=dP(i))
            ops.add_(d_v_prev, ops.gemm(dP(i), t['edge_gru_w_ih_t']))
            d_v, d_e = d_v_prev, d_e_prev
        # first calls (h = 0): gh = b_hh only
        ops.gru_gate_bwd(d_v, sv['gin'][0], None, imp.node_gru_b_hh, None, rows(dGIn, 0, N), rows(dGHn, 0, N), False)
        d_obj_rep = ops.gemm(rows(dGIn, 0, N), t['node_gru_w_ih_t'])
        ops.gru_gate_bwd(d_e, sv['gie0'], None, imp.edge_gru_b_hh, None, dgi_call(0), rows(dGHe, 0, E), False)
        d_rel_rep = ops.gemm(dgi_call(0), t['edge_gru_w_ih_t'])
        if T != 3:
            raise NotImplementedError('training is wired for mp_iter == 3')
        p = sv['dropout_p']
        ds = 1.0 / (1.0 - p) if p > 0 else 1.0
        C, PP = model.edge_dim, model.pool_sz ** 2
        _lib.set_tag('bwd_mlp')
        d_u = ops.act_bwd(d_rel_rep, XH[:E])                                       # relu(edge_unary)
        fc, edge_relu7 = model.fc_layers()
        n6e, n7e, n6o, n7o = (fc[k][0] for k in ('fc6_edge', 'fc7_edge', 'fc6_obj', 'fc7_obj'))
        d_y7 = lin_bwd(d_u, sv['y7'], t['edge_unary_t'], 'edge_unary', 'bwd_mlp')
        if edge_relu7:
            d_y7 = ops.act_bwd(d_y7, sv['y7'])
        d_y6 = lin_bwd(d_y7, sv['y6'], t['fc7_edge_t'], n7e, 'bwd_mlp', big=True)
        d_pre6 = ops.act_bwd(d_y6, sv['y6'], ds)                                   # dropout + relu
        _lib.set_tag('bwd_mlp_obj')
        d_x7 = lin_bwd(d_obj_rep, sv['x7'], t['obj_unary_t'], 'obj_unary', 'bwd_mlp_obj')
        d_p7 = ops.act_bwd(d_x7, sv['x7'], ds)
        d_x6 = lin_bwd(d_p7, sv['x6'], t['fc7_obj_t'], n7o, 'bwd_mlp_obj', big=True)
        d_p6 = ops.act_bwd(d_x6, sv['x6'], ds)
        # ---- the lane: everything below depends on phase A only and is latency- / HBM-bound (256-row contractions with 100 M outputs, the
        # node GRU's parameter gradients, the rect conv's backward with its two BatchNorms): it runs on the node lane's stream UNDER the
        # MFMA-bound fc6 / fc7 weight gradients of the edge branch instead of after them (SGG_BWD_LANE=0: in line, as before round 3)
        node_side = {'obj_fc', 'obj_unary', n7o}

        def lane_work():
            _lib.set_tag('bwd_mlp_obj')
            G[n6o + '.weight'], G[n6o + '.bias'] = tn_gemm(d_p6, sv['nf'], want_colsum=True, out_dtype=big_dtype())
            hook(n6o + '.weight')
            for name, dw in deferred[::-1]:                # fc7 node, unary node, obj_fc
                if name in node_side:
                    dw()
            _lib.set_tag('bwd_imp')
            G['node_gru.weight_ih'], G['node_gru.bias_ih'] = tn_gemm(dGIn, XN, want_colsum=True)
            G['node_gru.weight_hh'] = tn_gemm(dGHn[N:], HN[:T * N])
            G['node_gru.bias_hh'] = ops.colsum(dGHn)
            # HBM-bound reductions of the edge GRU / the gates (column sums over 4 E rows, rank-4 contractions): also under the GEMMs
            G['edge_gru.bias_ih'] = ops.colsum(DG[:(T + 1) * E])
            G['edge_gru.bias_hh'] = ops.colsum(dGHe)                          # b_hh acts in every call
            ops.rank4_reduce_(da_all, HE[:T * E], d_gw, col0=H, accumulate=False)         # e_i = rows i of HE, v_i = rows i of HN
            ops.rank4_reduce_(nsum_all, HN[:T * N], d_gw, col0=0, accumulate=False)
            d_gb = ops.colsum(da_all)                           # the gate biases: column sums of the gate pre-activation gradients
            for k, g in enumerate(GATES):                       # views of the two results (nothing writes them after this point)
                G[g + '.0.weight'] = d_gw[k:k + 1]
                G[g + '.0.bias'] = d_gb[k:k + 1]
            _lib.set_tag('bwd_rect')
            d_rect = ops.gemm(d_pre6, t['w6sum_t'])                                    # [E,C]
            # rect conv backward (BatchNorm with batch statistics)
            bn_sync = getattr(model, '_bn_sync', None)
            d_c2, db2, dg2 = ops.bn_bwd(d_rect, None, sv['h3'], sv['m2'], sv['is2'], t['rc_g2'], False, bn_sync)
            G['union_boxes.conv.6.weight'], G['union_boxes.conv.6.bias'] = dg2, db2
            gw2, gb2 = tn_gemm(d_c2, sv['h2'], want_colsum=True)                       # [d, d2] centre tap
            # only the centre tap of this 3x3 convolution sees data (1x1 input, padding 1): its gradient with a ring of zeros, one launch
            G['union_boxes.conv.4.weight'] = torch.nn.functional.pad(gw2[:, :, None, None], (1, 1, 1, 1))
            G['union_boxes.conv.4.bias'] = gb2
            d_h2 = ops.gemm(d_c2, t['rc_w2_t'])                                        # [E,d2]
            d_c1, db1, dg1 = ops.bn_bwd(d_h2, sv['arg'], sv['h1'], sv['m1'], sv['is1'], t['rc_g1'], True, bn_sync)
            G['union_boxes.conv.2.weight'], G['union_boxes.conv.2.bias'] = dg1, db1
            gw1, gb1 = tn_gemm(d_c1, sv['patches'], want_colsum=True)                  # [d2,128]
            G['union_boxes.conv.0.weight'] = gw1[:, :98].reshape(tuple(model.union_boxes.conv[0].weight.shape)).contiguous()
            G['union_boxes.conv.0.bias'] = gb1
            _lib.set_tag('')

        lane = node_lane(dev) if os.environ.get('SGG_BWD_LANE', '1') != '0' else None
        lane_done, lane_keys = None, ()
        # a capture in progress (sgg_amd/graph_step.py): the lane's work becomes a graph of its own, replayed on the lane's stream beside the
        # graph of phases B and C -- the capture is cut here, after the lane's work, and once more where the two meet
        split = getattr(model, '_graph_split', None)
        if split is not None:
            split.next('lane')
            lane_work()
            split.next('main')
        elif lane is not None:
            side = lane[0]
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(dev))
            side.wait_event(ready)                         # phase A is complete (d_p6, d_pre6, the node GRU's gate gradients)
            before = set(G)
            with torch.cuda.stream(side):
                lane_work()
                lane_done = torch.cuda.Event()
                lane_done.record(side)
            lane_keys = tuple(set(G) - before)
        # ---- phase B: the fc6 weight gradient of the edge branch
        _lib.set_tag('bwd_fc6_edge_dW')
        # d W6[n,(c,p)] = sum_e d_pre6[e,n] * (edge_feat[e,c,p] + rect[e,c]): the folded term rides in the transpose
        paired = sv['paired']
        x6t = d6t = None
        # x3 mode (strict backward): the pair-summed gradient rows and the pooled rows as PAIR operands through the TN form, both as they lie
        ef2d = sv['ef'].reshape(sv['ef'].shape[0], -1) if paired is not None else None
        x3_tn6 = (paired is not None and not sv.get('tn6') and 'x6t' not in sv and ops.split3_on() and not ops._BWD16[0] and USE_TN and
                  d_pre6.dtype == torch.float32 and ef2d.dtype == torch.float32 and ef2d.shape[0] % 32 == 0 and
                  ops.gemm_tn_x3_ok(ef2d.new_empty((ef2d.shape[0], d_pre6.shape[1])), ef2d))
        if x3_tn6:
            G[n6e + '.bias'] = ops.colsum(d_pre6)
            _lib.set_tag('bwd_fc6_rect_term')
            r6 = tn_gemm(d_pre6, sv['rect'])
            _lib.set_tag('bwd_fc6_edge_dW')
            G[n6e + '.weight'] = ops.group_bcast_add_(ops.gemm_tn_x3(ops.pairsum(d_pre6, paired.u2e), ef2d, out_dtype=big_dtype()), r6, PP)
        elif sv.get('tn6'):
            pass
        elif 'x6t' in sv:
            torch.cuda.current_stream(dev).wait_event(sv['x6t_ready'])
            x6t = sv['x6t']
        elif paired is not None:
            x6t = ops.transpose(sv['ef'])
        else:
            x6t = ops.transpose(sv['ef'], add=sv['rect'].float() if sv['rect'].dtype != torch.float32 else sv['rect'], group=PP)
        if x3_tn6:
            pass
        elif sv.get('tn6'):
            # the same two terms as below, the first as  (pair sums of d_pre6)^T . pooled  with both operands read as they lie
            G[n6e + '.bias'] = ops.colsum(d_pre6)
            _lib.set_tag('bwd_fc6_rect_term')
            r6 = tn_gemm(d_pre6, sv['rect'])
            _lib.set_tag('bwd_fc6_edge_dW')
            G[n6e + '.weight'] = ops.gemm_tn_full_waves(ops.pairsum(d_pre6, paired.u2e), sv['ef'], out_dtype=big_dtype(), gadd=(r6, PP))
        elif paired is not None:
            # rows of the unordered pairs: d W6[n,(c,p)] = sum_u (d_pre6[e1(u),n] + d_pre6[e2(u),n]) pooled[u,c,p]  +  (sum_e d_pre6[e,n] rect[e,c])
            # broadcast over p -- the second term is the gradient through W6's group sums (the folded rect term of the forward)
            d6t = ops.transpose_pairsum(d_pre6, paired.u2e)
            G[n6e + '.bias'] = ops.colsum(d_pre6)
            G[n6e + '.weight'] = ops.gemm_full_waves(d6t, x6t, out_dtype=big_dtype(), gadd=(tn_gemm(d_pre6, sv['rect']), PP))
        else:
            d6t, G[n6e + '.bias'] = ops.transpose(d_pre6, want_colsum=True)
            G[n6e + '.weight'] = ops.gemm_full_waves(d6t, x6t, out_dtype=big_dtype())
        hook(n6e + '.weight')
        del x6t, d6t
        # SGG_GRAPH_EARLY_JOIN=1 (replayed step): the lane's graph meets the calling stream HERE, right after the fc6 weight gradient, and phase C
        # runs with no kernel of another queue resident.  Off by default since the end of round 5: stream-position stamps (tools/step_stamps.py)
        # show the calling stream idle for ~0.2 ms at that meeting point while the lane finishes (its work is stretched ~1.8x under the long
        # launch), and phase C beside the lane's tail costs less than that wait -- same box, alternating: 7.20 / 7.09 / 7.20 / 7.05 / 7.20 / 7.06 ms per
        # bench step (profiles/r05_join_placement.txt); the earlier 6.79 -> 6.755 ms in favour of the early join was inside the noise
        early_join = split is not None and os.environ.get('SGG_GRAPH_EARLY_JOIN', '0') == '1'
        if early_join:
            split.next('joined')
        # ---- phase C on the main stream: the edge-side weight gradients, largest first (fc7 carries its own hook)
        for name, dw in deferred[::-1]:                # fc7 edge, unary edge, rel_fc (the node-side ones ran / run on the lane)
            if name not in node_side:
                dw()
        _lib.set_tag('bwd_imp')
        # GRU parameter gradients: one contraction over the 4 stacked calls; hidden state of call 0 is zero
        G['edge_gru.weight_ih'] = tn_gemm(DG[T * E:], XH[:E + T * N])           # [d_gi of call 0 ; dP_0 ..]^T . [rel_rep ; v_0 ..]
        G['edge_gru.weight_hh'] = tn_gemm(dGHe[E:], HE[:T * E])              # states entering calls 1..T (call 0: zero state)
        if split is not None:
            if not early_join:
                split.next('joined')                        # what follows (and the memory it recycles) runs after the lane's graph has finished
        elif lane_done is None:
            lane_work()
        else:                                               # everything the lane produced is complete before anything downstream reads it
            main = torch.cuda.current_stream(dev)
            main.wait_event(lane_done)
            for kname in lane_keys:
                if isinstance(G.get(kname), torch.Tensor):
                    G[kname].record_stream(main)          # allocated under the lane's stream, consumed (and freed) under this one
        _lib.set_tag('')
        ctx.sv = None
        shapes = dict(model.head_named_parameters())
        # gradients already handed to the all-reduce in the wire dtype are not returned to autograd (no fp32 copy)
        grads = [None if n in handed else G[n].reshape(shapes[n].shape) for n in param_names(model)]
        # gradients into the RoI features, only where a caller keeps them in the graph (GAN feature augmentation, main.py:145-149:
        # predict() on generated features): dX of the two fc6 layers, [rows, 4096] . W6 -> [rows, C*P*P] in the reference's (c,ph,pw) order
        d_nf = d_ef = None
        if ctx.needs_input_grad[1]:
            d_nf = ops.gemm(d_p6, ops.transpose(w['fc6_obj']), out_dtype=torch.float32)
        if ctx.needs_input_grad[2]:
            if paired is not None:
                raise NotImplementedError('gradient into pair-pooled edge features (pass dense edge features to predict())')
            d_ef = ops.gemm(d_pre6, ops.transpose(w['fc6_edge']), out_dtype=torch.float32)
        return (None, d_nf, d_ef) + (None,) * 5 + tuple(grads)


def predict_train(model, node_feat, edge_feat, rel_inds, rois, im_inds=None, seed=None, dropout_p=DROPOUT_P, graphs=None,
                  im_sizes=None, pairing=None, csr=None):
    """Autograd-connected training forward of the head.  node_feat/edge_feat: [.,P,P,C]-contiguous (NHWC) tensors
    in the compute dtype."""
    N, E = node_feat.shape[0], edge_feat.shape[0]      # (E: rows of edge_feat -- the unordered pairs when `pairing` is given)
    if seed is None:
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    named = dict(model.head_named_parameters())
    params = [named[n] for n in param_names(model)]
    model._graphs_hint = graphs     # host-side facts about the graphs (ops.edge_csr), read by PredictFn.forward
    model._im_sizes_hint = im_sizes  # image sizes for the 'raw_boxes' raster (lib/get_union_boxes.py:71-78)
    model._pairing_hint = pairing    # sgg_amd/pairing.py: edge_feat holds one row per unordered box pair
    model._csr_hint = csr
    return PredictFn.apply(model, node_feat.reshape(N, -1), edge_feat.reshape(E, -1), rois.float().contiguous(),
                           rel_inds.contiguous(), im_inds, seed, float(dropout_p), *params)

