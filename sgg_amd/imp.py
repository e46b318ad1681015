"""Iterative message passing on the HIP path: RelModelStanford.message_pass, sgg_models/rel_model_stanford.py:48-94."""
import torch

from . import ops

GATES = ('sub_vert_w_fc', 'obj_vert_w_fc', 'out_edge_w_fc', 'in_edge_w_fc')  # rel_model_stanford.py:41-45


class ImpWeights(object):
    """Device operands of the loop: GRU matrices in the compute dtype, biases and the four gate vectors in fp32."""

    @classmethod
    def from_state(cls, p, dtype, cast=None):
        """p: {name: parameter}.  cast(name, parameter) -> the compute-dtype operand (the model passes its shadow buffers, which a
        fused optimiser keeps current in its update pass; default: a cast).  The four gate layers (Linear(2H, 1) each) are kept in
        ONE [4,2H] / [4] fp32 block: their parameters are re-pointed at views of it (once; again after something gave them storage
        of their own, e.g. module.to()), so the kernels read the block the optimiser updates -- no per-step concatenation."""
        w = cls()
        cast = cast or (lambda name, t: t.detach().to(dtype).contiguous())
        for g in ('edge_gru', 'node_gru'):
            setattr(w, g + '_w_ih', cast(g + '.weight_ih', p[g + '.weight_ih']))
            setattr(w, g + '_w_hh', cast(g + '.weight_hh', p[g + '.weight_hh']))
            setattr(w, g + '_b_ih', p[g + '.bias_ih'].detach().float().contiguous())
            setattr(w, g + '_b_hh', p[g + '.bias_hh'].detach().float().contiguous())
        w.gate_w = pack_rows([p[g + '.0.weight'] for g in GATES])              # [4,2H]: (vertex half | edge half) of each gate
        w.gate_b = pack_rows([p[g + '.0.bias'] for g in GATES]).view(-1)       # [4]
        w.H = w.edge_gru_w_hh.shape[1]
        return w


def pack_rows(params):
    """-> f32 [len(params), n] whose row i IS params[i] (each a parameter of n elements): the parameters' storage is moved into one
    block the first time (and whenever they are found apart again); afterwards this costs no launch."""
    n = params[0].numel()
    first = params[0].data
    esz = first.element_size()
    packed = (first.dtype == torch.float32 and first.is_contiguous() and
              all(q.dtype == torch.float32 and q.numel() == n and q.data.is_contiguous() and
                  q.data.untyped_storage().data_ptr() == first.untyped_storage().data_ptr() and
                  q.data_ptr() == first.data_ptr() + i * n * esz for i, q in enumerate(params)) and
              first.untyped_storage().nbytes() >= first.storage_offset() * esz + len(params) * n * esz)
    if not packed:
        with torch.no_grad():
            block = torch.cat([q.detach().float().reshape(1, n) for q in params], 0).contiguous()
            for i, q in enumerate(params):
                q.data = block[i].view(q.shape)
        first = params[0].data
    return torch.as_strided(first, (len(params), n), (n, 1))


def _gru(wts, which, x, h, dtype, out=None, dots=False):
    """nn.GRUCell: two MFMA GEMMs (fp32 pre-activations) + the fused pointwise gate kernel.  h is None for the
    first call (hidden state 0: the hidden GEMM is skipped, b_hh still applies -- rel_model_stanford.py:68-72).
    dots=True: also the four gate dot products of the new state (vertex halves for node_gru, edge halves for edge_gru),
    -> (state, dots f32[M,4])."""
    gi = ops.gemm(x, getattr(wts, which + '_w_ih'), getattr(wts, which + '_b_ih'), out_dtype=torch.float32)
    dot_w = (wts.gate_w[:, :wts.H] if which == 'node_gru' else wts.gate_w[:, wts.H:]) if dots else None
    if h is None:
        return ops.gru_gate(gi, None, getattr(wts, which + '_b_hh'), None, dtype, out=out, dot_w=dot_w)
    gh = ops.gemm(h, getattr(wts, which + '_w_hh'), getattr(wts, which + '_b_hh'), out_dtype=torch.float32)
    return ops.gru_gate(gi, gh, None, h, dtype, out=out, dot_w=dot_w)


_SIDE = {}


def node_lane(device):
    """A second HIP stream (+ two events) per device for the node-side work of the loop.  The node GRU is three tiny
    GEMMs and a gate kernel on 32B rows -- pure launch/DRAM latency (~15 us each) -- and depends only on (ctx, v_i); run
    on its own stream it hides behind the edge GRU of the same iteration instead of queueing after it.
    SGG_IMP_STREAMS=0 keeps everything on one stream."""
    import os
    if os.environ.get('SGG_IMP_STREAMS', '1') == '0':
        return None
    key = (device.type, device.index)
    if key not in _SIDE:
        _SIDE[key] = (torch.cuda.Stream(device=device), torch.cuda.Event(), torch.cuda.Event())
    return _SIDE[key]


def message_pass(rel_rep, obj_rep, rel_inds, csr, wts, mp_iter, dtype):
    """rel_rep [E,H], obj_rep [N,H] (dtype), rel_inds i64[E,3] -> (vert[N,H], edge[E,H]).  rel_model_stanford.py:68-94 with the node
    projection (csrc/imp.hip): per iteration
        ctx   = imp_ctx(e_i)                                   :86-91  the step's read stream (any edge list; sorted graphs: every row once)
        P_i   = v_i W_ih^T                                     [N,3H] f32 -- replaces e_in and its [E,3H] projection (:76-83)
        e_i+1 = gru_gate_proj(e_i W_hh^T + b_hh, P_i[s], P_i[o], gates)
        v_i+1 = GRU_n(ctx, v_i)                                :92
    The gate pre-activations travel as dot products emitted by the kernels that write v and e."""
    N, H = obj_rep.shape
    if mp_iter > 0 and not ops.gate_dots_ok(H):
        raise NotImplementedError('message passing needs hidden_dim / 8 to be a power of two <= 64 (got hidden_dim %d)' % H)
    lane = node_lane(obj_rep.device)
    loop = mp_iter > 0
    wv = wts.gate_w[:, :H]

    def unpack(r, want):
        return r if want else (r, None)

    def node_step(ctx2, vert, more, out=None, dots=None):
        # :92  node_gru(ctx_out + ctx_in, vert): the sum rides in the GEMM's K axis
        gi = ops.gemm(ctx2[0], wts.node_gru_w_ih, wts.node_gru_b_ih, out_dtype=torch.float32, A2=ctx2[1], W2=wts.node_gru_w_ih)
        gh = ops.gemm(vert, wts.node_gru_w_hh, wts.node_gru_b_hh, out_dtype=torch.float32)
        return unpack(ops.gru_gate(gi, gh, None, vert, dtype, out=out, dot_w=wv if more else None, dots=dots), more)

    def edge_step(edge, P, nd, ed, more):
        gh = ops.gemm(edge, wts.edge_gru_w_hh, wts.edge_gru_b_hh, out_dtype=ops.gh_dtype(dtype))
        return unpack(ops.gru_gate_proj(gh, P, wts.edge_gru_b_ih, csr, nd, ed, wts.gate_b, edge, dot_w=wts.gate_w[:, H:] if more else None), more)

    if lane is None:
        vert, nd = unpack(_gru(wts, 'node_gru', obj_rep, None, dtype, dots=loop), loop)       # :71
        edge, ed = unpack(_gru(wts, 'edge_gru', rel_rep, None, dtype, dots=loop), loop)       # :72
        for i in range(mp_iter):                                                 # :74
            more = i + 1 < mp_iter
            ctx2 = ops.imp_ctx(edge, csr, N, nd, ed, wts.gate_b)
            P = ops.gemm(vert, wts.edge_gru_w_ih, None, out_dtype=torch.float32)
            edge_new, ed_new = edge_step(edge, P, nd, ed, more)                  # :83
            vert, nd = node_step(ctx2, vert, more)
            edge, ed = edge_new, ed_new
        return vert, edge
    # Two streams.  Memory rules that keep the caching allocator out of trouble: every tensor that crosses streams is
    # allocated on the main stream and stays referenced until the main stream has waited for the last side-stream
    # event (verts, ndots, Ps, keep); temporaries of the side stream (gi, gh) are allocated and freed under the side stream.
    side, ev_main, ev_side = lane
    dev = obj_rep.device
    main = torch.cuda.current_stream(dev)
    verts = [torch.empty((N, H), dtype=dtype, device=dev) for _ in range(mp_iter + 1)]
    ndots = [torch.empty((N, 4), dtype=torch.float32, device=dev) for _ in range(mp_iter)]
    Ps = [torch.empty((N, 3 * H), dtype=torch.float32, device=dev) for _ in range(mp_iter)]
    keep = []
    ev_main.record(main)
    side.wait_event(ev_main)                                                     # obj_rep is ready
    with torch.cuda.stream(side):
        gi = ops.gemm(obj_rep, wts.node_gru_w_ih, wts.node_gru_b_ih, out_dtype=torch.float32)
        ops.gru_gate(gi, None, wts.node_gru_b_hh, None, dtype, out=verts[0],                 # :71
                     dot_w=wv if loop else None, dots=ndots[0] if loop else None)
        del gi
        if loop:
            ops.gemm(verts[0], wts.edge_gru_w_ih, None, out_dtype=torch.float32, out=Ps[0])
        ev_side.record(side)
    edge, ed = unpack(_gru(wts, 'edge_gru', rel_rep, None, dtype, dots=loop), loop)           # :72
    for i in range(mp_iter):                                                     # :74
        more = i + 1 < mp_iter
        main.wait_event(ev_side)                                                 # v_i, its gate dots and its projection are ready
        ctx2 = ops.imp_ctx(edge, csr, N, ndots[i], ed, wts.gate_b)
        keep.append(ctx2)
        ev_main.record(main)
        side.wait_event(ev_main)
        with torch.cuda.stream(side):
            node_step(ctx2, verts[i], more, out=verts[i + 1], dots=ndots[i + 1] if more else None)
            if more:
                ops.gemm(verts[i + 1], wts.edge_gru_w_ih, None, out_dtype=torch.float32, out=Ps[i + 1])
            ev_side.record(side)
        edge, ed = edge_step(edge, Ps[i], ndots[i], ed, more)                    # :83
    main.wait_event(ev_side)
    return verts[-1], edge
