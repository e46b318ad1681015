"""Where does the bf16 mode's logit error come from?  (VERDICT r2 "What's weak" #1: obj max 0.41 / rel max 0.10 against a 1e-3 clause.)

Runs the bench batch (8 x 592x592, 32 boxes, 992 edges, seed 111) through the real kernels in fp32 and bf16 up to the RoI features, and
through a torch emulation of the head (fc6 / fc7 / unary / IMP / heads: f32 matmuls whose operands and stored activations are rounded to
bf16 where a switch says so) from there on.  Every line = one set of rounding points; the number = max / mean |logit - fp32 logit|.
    python tools/precision_probe.py            (GPU; about a minute)
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sgg_amd  # noqa: E402
from sgg_amd import ops  # noqa: E402
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch  # noqa: E402

DEV = 'cuda:0'
B = int(os.environ.get('B', '8'))


FMT = [torch.bfloat16]


def r16(x, on=True):
    """storage rounding: f32 -> 16-bit format (FMT[0]) -> f32"""
    return x.to(FMT[0]).float() if on else x


def vgg_emulated(model, x_nchw, on=True):
    """VGG-16 features (last pool dropped) with every conv's operands and stored output rounded to the 16-bit format."""
    import torch.nn.functional as F
    x = r16(x_nchw, on)
    for m in model.detector.backbone:
        if isinstance(m, torch.nn.Conv2d):
            x = r16(torch.relu(F.conv2d(x, r16(m.weight.detach().float(), on), m.bias.detach().float(), padding=1)), on)
        elif isinstance(m, torch.nn.MaxPool2d):
            x = F.max_pool2d(x, 2)
    return x


def mm(x, w, b=None, op16=True):
    """x [M,K] . w[N,K]^T + b with f32 accumulation; op16: both operands rounded to bf16 (what the MFMA sees)."""
    y = r16(x, op16) @ r16(w, op16).t()
    return y + b if b is not None else y


def features(model, batch, dtype):
    model.set_compute_dtype(dtype)
    with torch.no_grad():
        res = model.faster_rcnn(batch[0], batch[3].to(DEV), batch[4].to(DEV), None)
        rel_inds = model.get_rel_inds(None, res.im_inds, res.rm_box_priors)
        rois = torch.cat((res.im_inds[:, None].float(), res.rm_box_priors), 1)
        fmap = res.fmap.float().clone()
        nf, ef = model.node_edge_features(res.fmap, rois, rel_inds[:, 1:], res.im_sizes)
        rect = model.union_boxes.rect_feat(rois, rel_inds[:, 1:].contiguous(), dtype, res.im_sizes)
    return dict(fmap=fmap, nf=nf.float().reshape(nf.shape[0], -1), ef=ef.float().reshape(ef.shape[0], -1), rect=rect.float(),
                rel_inds=rel_inds, rois=rois, im_sizes=res.im_sizes)


def features_from_fmap(model, fmap, rois, rel_inds, im_sizes, dtype):
    model.set_compute_dtype(dtype)
    with torch.no_grad():
        nf, ef = model.node_edge_features(fmap.to(dtype) if dtype != torch.float32 else fmap, rois, rel_inds[:, 1:], im_sizes)
    return nf.float().reshape(nf.shape[0], -1), ef.float().reshape(ef.shape[0], -1)


def head(P, nf, ef, rect, rel_inds, s):
    """Emulated predict() (rel_model_stanford.py:97-107).  s = dict of switches (True = rounded to bf16 there):
    in_n / in_e: RoI features stored bf16; w_n / w_e: node / edge MLP GEMM operands bf16; act_n / act_e: MLP activations stored bf16;
    gru_op_n / gru_op_e: GRU GEMM operands bf16; st_n / st_e: GRU states stored bf16; ctx: ctx stored bf16; ein: e_in stored bf16;
    head_n / head_e: head GEMM operands bf16."""
    g = lambda k: s.get(k, False)
    E = ef.shape[0]
    # nodes
    x = r16(nf, g('in_n'))
    x = r16(torch.relu(mm(x, P['roi_fmap_obj.0.weight'], P['roi_fmap_obj.0.bias'], g('w_n'))), g('act_n'))
    x = r16(torch.relu(mm(x, P['roi_fmap_obj.3.weight'], P['roi_fmap_obj.3.bias'], g('w_n'))), g('act_n'))
    obj_rep = r16(mm(x, P['obj_unary.weight'], P['obj_unary.bias'], g('w_n')), g('act_n'))
    # edges: fc6(x + rect (x) 1_49) = fc6(x) + W6sum rect
    W6 = P['roi_fmap.1.0.weight']
    W6sum = W6.view(W6.shape[0], -1, 49).sum(2)
    y = torch.empty((E, W6.shape[0]), device=ef.device)
    for a in range(0, E, 1024):
        y[a:a + 1024] = mm(r16(ef[a:a + 1024], g('in_e')), W6, None, g('w_e'))
    y = y + mm(r16(rect, g('in_e')), W6sum, P['roi_fmap.1.0.bias'], g('w_e'))
    y = r16(torch.relu(y), g('act_e'))
    y = r16(mm(y, P['roi_fmap.1.3.weight'], P['roi_fmap.1.3.bias'], g('w_e')), g('act_e'))
    rel_rep = r16(torch.relu(mm(y, P['edge_unary.weight'], P['edge_unary.bias'], g('w_e'))), g('act_e'))
    if 'inject' in s:
        obj_rep, rel_rep = s['inject'](obj_rep, rel_rep)
    return imp_heads(P, obj_rep, rel_rep, rel_inds, s) + (obj_rep, rel_rep)


def gru(P, name, x, h, op16):
    gi = mm(x, P[name + '.weight_ih'], P[name + '.bias_ih'], op16)
    gh = mm(h, P[name + '.weight_hh'], P[name + '.bias_hh'], op16) if h is not None else P[name + '.bias_hh'][None, :].expand_as(gi)
    H = gi.shape[1] // 3
    r = torch.sigmoid(gi[:, :H] + gh[:, :H])
    z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1 - z) * n + z * (h if h is not None else 0)


def imp_heads(P, obj_rep, rel_rep, rel_inds, s):
    g = lambda k: s.get(k, False)
    so = rel_inds[:, 1:]
    sub, ob = so[:, 0], so[:, 1]
    H = obj_rep.shape[1]
    gw = [P[k + '.0.weight'].view(-1) for k in ('sub_vert_w_fc', 'obj_vert_w_fc', 'out_edge_w_fc', 'in_edge_w_fc')]
    gb = [P[k + '.0.bias'].view(-1) for k in ('sub_vert_w_fc', 'obj_vert_w_fc', 'out_edge_w_fc', 'in_edge_w_fc')]
    v = r16(gru(P, 'node_gru', obj_rep, None, g('gru_op_n')), g('st_n'))
    e = r16(gru(P, 'edge_gru', rel_rep, None, g('gru_op_e')), g('st_e'))
    for _ in range(3):
        sv, ov = v[sub], v[ob]
        gate = lambda k, vert: torch.sigmoid(vert @ gw[k][:H] + e @ gw[k][H:] + gb[k])[:, None]
        e_in = r16(gate(0, sv) * sv + gate(1, ov) * ov, g('ein'))
        ctx = torch.zeros_like(v)
        if g('ctx_halves'):   # the two halves stored separately in bf16, summed in the GEMM's K axis
            c1 = torch.zeros_like(v).index_add_(0, sub, gate(2, sv) * e)
            c2 = torch.zeros_like(v).index_add_(0, ob, gate(3, ov) * e)
            ctxs = (r16(c1, g('ctx')), r16(c2, g('ctx')))
            Wih = P['node_gru.weight_ih']
            gi_extra = mm(ctxs[0], Wih, None, g('gru_op_n')) + mm(ctxs[1], Wih, None, g('gru_op_n'))
            ctx = None
        else:
            ctx = r16(ctx.index_add_(0, sub, gate(2, sv) * e).index_add_(0, ob, gate(3, ov) * e), g('ctx'))
        e_new = r16(gru(P, 'edge_gru', e_in, e, g('gru_op_e')), g('st_e'))
        if ctx is None:
            name = 'node_gru'
            gi = gi_extra + P[name + '.bias_ih']
            gh = mm(v, P[name + '.weight_hh'], P[name + '.bias_hh'], g('gru_op_n'))
            r = torch.sigmoid(gi[:, :H] + gh[:, :H]); z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
            n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
            v_new = (1 - z) * n + z * v
        else:
            v_new = gru(P, 'node_gru', ctx, v, g('gru_op_n'))
        v, e = r16(v_new, g('st_n')), e_new
    od = mm(v, P['obj_fc.weight'], P['obj_fc.bias'], g('head_n'))
    rd = mm(e, P['rel_fc.weight'], P['rel_fc.bias'], g('head_e'))
    return od, rd


def main():
    torch.manual_seed(0)
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(DEV).eval()
    P = {k: v.detach().float() for k, v in model.state_dict().items() if not k.startswith('detector.')}
    batch = synthetic_batch(B=B, S=592, n_boxes=32, n_fg=6, seed=111)
    os.environ['SGG_EDGE_PAIRS'] = '0'
    F32 = features(model, batch, torch.float32)
    F16 = features(model, batch, torch.bfloat16)
    ri = F32['rel_inds']
    nf_x, ef_x = features_from_fmap(model, F32['fmap'], F32['rois'], ri, F32['im_sizes'], torch.bfloat16)   # exact fmap -> bf16 RoIAlign
    ALL = dict(in_n=1, in_e=1, w_n=1, w_e=1, act_n=1, act_e=1, gru_op_n=1, gru_op_e=1, st_n=1, st_e=1, ctx=1, ein=1, head_n=1, head_e=1,
               ctx_halves=1)
    with torch.no_grad():
        od0, rd0, orep0, rrep0 = head(P, F32['nf'], F32['ef'], F32['rect'], ri, {})
        # the real kernels, both modes
        real = {}
        for name, dt in (('f32', torch.float32), ('bf16', torch.bfloat16)):
            model.set_compute_dtype(dt)
            res = model.faster_rcnn(batch[0], batch[3].to(DEV), batch[4].to(DEV), None)
            rois = torch.cat((res.im_inds[:, None].float(), res.rm_box_priors), 1)
            nf, ef = model.node_edge_features(res.fmap, rois, ri[:, 1:], res.im_sizes)
            od, rd = model.predict(nf, ef, ri, rois, res.im_sizes, _im_inds=res.im_inds.contiguous())
            real[name] = (od.float(), rd.float())
    rows = []

    def report(label, od, rd):
        eo, er = (od - od0).abs(), (rd - rd0).abs()
        row = dict(label=label, obj_max=float(eo.max()), obj_mean=float(eo.mean()), rel_max=float(er.max()), rel_mean=float(er.mean()))
        rows.append(row)
        print('%-74s obj max %.4f mean %.5f | rel max %.4f mean %.5f' % (label, row['obj_max'], row['obj_mean'], row['rel_max'], row['rel_mean']),
              flush=True)

    print('fmap |bf16 - f32| max %.4f mean %.5f (|f32| max %.2f mean %.4f)' % (
        float((F16['fmap'] - F32['fmap']).abs().max()), float((F16['fmap'] - F32['fmap']).abs().mean()),
        float(F32['fmap'].abs().max()), float(F32['fmap'].abs().mean())))
    print('logit magnitudes: obj max %.2f, rel max %.2f' % (float(od0.abs().max()), float(rd0.abs().max())))
    report('REAL kernels f32 mode vs emulated f32', *real['f32'])
    report('REAL kernels bf16 mode', *real['bf16'])
    with torch.no_grad():
        run = lambda nf, ef, rect, s: head(P, nf, ef, rect, ri, s)[:2]
        report('emulated: everything bf16 (features from the bf16 kernels)', *run(F16['nf'], F16['ef'], F16['rect'], ALL))
        report('emulated: bf16 VGG+RoIAlign features, head exact', *run(F16['nf'], F16['ef'], F16['rect'], {}))
        report('emulated: exact fmap -> bf16 RoIAlign storage, head exact', *run(nf_x, ef_x, F32['rect'], {}))
        report('emulated: exact features, head all bf16', *run(F32['nf'], F32['ef'], F32['rect'], ALL))
        report('emulated: exact features, only RoI feature storage bf16 (in_n,in_e)', *run(F32['nf'], F32['ef'], F32['rect'], dict(in_n=1, in_e=1)))
        report('emulated: exact features, only MLP operands+acts bf16 (node)', *run(F32['nf'], F32['ef'], F32['rect'], dict(in_n=1, w_n=1, act_n=1)))
        report('emulated: exact features, only MLP operands+acts bf16 (edge)', *run(F32['nf'], F32['ef'], F32['rect'], dict(in_e=1, w_e=1, act_e=1)))
        imp_all = dict(gru_op_n=1, gru_op_e=1, st_n=1, st_e=1, ctx=1, ein=1, head_n=1, head_e=1, ctx_halves=1)
        report('emulated: exact MLP, IMP + heads all bf16', *run(F32['nf'], F32['ef'], F32['rect'], imp_all))
        for k in ('gru_op_n', 'gru_op_e', 'st_n', 'st_e', 'ctx', 'ein', 'head_n', 'head_e'):
            report('emulated: exact MLP, IMP only %s' % k, *run(F32['nf'], F32['ef'], F32['rect'], {k: 1, 'ctx_halves': 1}))
        node32 = dict(ALL)
        for k in ('st_n', 'ctx', 'gru_op_n', 'head_n'):
            node32.pop(k)
        report('emulated: all bf16 BUT node side of IMP+head fp32 (st_n, ctx, gru_op_n, head_n)', *run(F16['nf'], F16['ef'], F16['rect'], node32))
        node32b = dict(node32)
        for k in ('w_n', 'act_n'):
            node32b.pop(k)
        report('emulated:   ... and node MLP operands / activations fp32 too', *run(F16['nf'], F16['ef'], F16['rect'], node32b))
        e32 = dict(node32b)
        for k in ('st_e', 'ein', 'head_e'):
            e32.pop(k)
        report('emulated:   ... and edge states / e_in / rel head fp32', *run(F16['nf'], F16['ef'], F16['rect'], e32))
        e32b = dict(e32)
        e32b.pop('gru_op_e')
        report('emulated:   ... and edge GRU operands fp32', *run(F16['nf'], F16['ef'], F16['rect'], e32b))
        e32c = dict(e32b)
        e32c.pop('act_e')
        report('emulated:   ... and edge MLP activations fp32 (operands still bf16)', *run(F16['nf'], F16['ef'], F16['rect'], e32c))
        report('emulated: all bf16 with exact-fmap features', *run(nf_x, ef_x, F32['rect'], ALL))
        # ---- the whole forward emulated in a 16-bit format, VGG included: bf16 (cross-check against the real kernels) and fp16
        img = [b for k, b in model.detector._bufs.items() if k[0] == 'img'][0]
        x_nchw = img[:, 1:-1, 1:-1, :3].permute(0, 3, 1, 2).contiguous()
        fm32 = vgg_emulated(model, x_nchw, False)
        print('emulated f32 VGG vs real f32 fmap: max %.2e' % float((fm32 - F32['fmap']).abs().max()))
        for fmt in (torch.bfloat16, torch.float16):
            FMT[0] = fmt
            fm = vgg_emulated(model, x_nchw, True)
            d = (fm - F32['fmap']).abs()
            print('%s: emulated fmap error max %.4f mean %.5f; fmap absmax %.1f' % (fmt, float(d.max()), float(d.mean()), float(fm.abs().max())))
            model.set_compute_dtype(torch.float32)
            nf_e, ef_e = model.node_edge_features(fm, F32['rois'], ri[:, 1:], F32['im_sizes'])
            nf_e, ef_e = r16(nf_e.float().reshape(nf_e.shape[0], -1)), r16(ef_e.float().reshape(ef_e.shape[0], -1))
            report('emulated %s END TO END (VGG, RoIAlign storage, head)' % fmt, *run(nf_e, ef_e, r16(F32['rect']), ALL))
            report('emulated %s VGG + RoIAlign storage, head exact' % fmt, *run(nf_e, ef_e, F32['rect'], {}))
            report('emulated %s head only (exact features)' % fmt, *run(F32['nf'], F32['ef'], F32['rect'], ALL))
        FMT[0] = torch.bfloat16
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'precision_probe.json'), 'w') as f:
        json.dump(rows, f, indent=1)


if __name__ == '__main__':
    main()
