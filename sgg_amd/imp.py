"""Iterative message passing on the HIP path: RelModelStanford.message_pass, sgg_models/rel_model_stanford.py:48-94."""
import torch

from . import ops

GATES = ('sub_vert_w_fc', 'obj_vert_w_fc', 'out_edge_w_fc', 'in_edge_w_fc')  # rel_model_stanford.py:41-45


class ImpWeights(object):
    """Device operands of the loop: GRU matrices in the compute dtype, biases and the four gate vectors in fp32."""

    @classmethod
    def from_state(cls, p, dtype):
        w = cls()
        for g in ('edge_gru', 'node_gru'):
            setattr(w, g + '_w_ih', p[g + '.weight_ih'].detach().to(dtype).contiguous())
            setattr(w, g + '_w_hh', p[g + '.weight_hh'].detach().to(dtype).contiguous())
            setattr(w, g + '_b_ih', p[g + '.bias_ih'].detach().float().contiguous())
            setattr(w, g + '_b_hh', p[g + '.bias_hh'].detach().float().contiguous())
        w.gate_w = torch.cat([p[g + '.0.weight'].detach().float().reshape(1, -1) for g in GATES], 0).contiguous()
        w.gate_w_c = w.gate_w.to(dtype).contiguous()      # compute-dtype copy for the fused kernel
        w.gate_b = torch.cat([p[g + '.0.bias'].detach().float().reshape(1) for g in GATES], 0).contiguous()
        w.H = w.edge_gru_w_hh.shape[1]
        # node GRU input weight doubled along K: ctx = ctx_out + ctx_in is fed as a K-split operand (linearity)
        w.node_gru_w_ih2 = torch.cat((w.node_gru_w_ih, w.node_gru_w_ih), 1).contiguous()
        return w


def _gru(wts, which, x, h, dtype):
    """nn.GRUCell: two MFMA GEMMs (fp32 pre-activations) + the fused pointwise gate kernel.  h is None for the
    first call (hidden state 0: the hidden GEMM is skipped, b_hh still applies -- rel_model_stanford.py:68-72)."""
    gi = ops.gemm(x, getattr(wts, which + '_w_ih'), getattr(wts, which + '_b_ih'), out_dtype=torch.float32)
    if h is None:
        return ops.gru_gate(gi, None, getattr(wts, which + '_b_hh'), None, dtype)
    gh = ops.gemm(h, getattr(wts, which + '_w_hh'), getattr(wts, which + '_b_hh'), out_dtype=torch.float32)
    return ops.gru_gate(gi, gh, None, h, dtype)


def message_pass(rel_rep, obj_rep, rel_inds, csr, wts, mp_iter, dtype):
    """rel_rep [E,H], obj_rep [N,H] (dtype), rel_inds i64[E,3] -> (vert[N,H], edge[E,H])."""
    N = obj_rep.shape[0]
    vert = _gru(wts, 'node_gru', obj_rep, None, dtype)                       # :71
    edge = _gru(wts, 'edge_gru', rel_rep, None, dtype)                       # :72
    for _ in range(mp_iter):                                                 # :74
        e_in, ctx2 = ops.imp_fused(vert, edge, rel_inds, csr, wts.gate_w_c, wts.gate_b)   # :76-81,86-91 in one launch
        edge_new = _gru(wts, 'edge_gru', e_in, edge, dtype)                  # :83
        # :92  node_gru(ctx_out + ctx_in, vert): the sum rides in the GEMM's K axis
        gi = ops.gemm(ctx2[0], wts.node_gru_w_ih2, wts.node_gru_b_ih, out_dtype=torch.float32, A2=ctx2[1])
        gh = ops.gemm(vert, wts.node_gru_w_hh, wts.node_gru_b_hh, out_dtype=torch.float32)
        vert = ops.gru_gate(gi, gh, None, vert, dtype)
        edge = edge_new
    return vert, edge
