"""Shared by tests/test_parity_full_gpu.py and tests/test_zz_x3_bench_gpu.py: the BENCHMARKED configuration (8 x 592x592 frames, 32 boxes and
992 edges each, seed 111 -- bench.py's rank-0 batch) through the fp32 CPU oracle ONCE per process, and through the HIP forward in a named mode.
Every mode has its own test (one failing assertion stopped the driver's `-x` run in front of 90 tests)."""
import json
import os
import time

import numpy as np
import torch

from oracle import sgg_oracle as O

DEV = 'cuda:0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_STATE = {}


def dump(name, payload, merge=False):
    out = os.path.join(ROOT, 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, name)
        if merge and os.path.exists(path):
            with open(path) as f:
                old = json.load(f)
            old.update(payload)
            payload = old
        with open(path, 'w') as f:
            json.dump(payload, f, indent=1, sort_keys=True)
    except (OSError, ValueError):
        pass


def logits(model, batch):
    """(obj_dists, rel_dists, rel_inds) of the eval forward, before the tail."""
    dev = model.rel_fc.weight.device
    with torch.no_grad():
        res = model.faster_rcnn(batch[0], batch[3].to(dev), batch[4].to(dev), None)
        rel_inds = model.get_rel_inds(None, res.im_inds, res.rm_box_priors)
        rois = torch.cat((res.im_inds[:, None].float(), res.rm_box_priors), 1)
        nf, ef = model.node_edge_features(res.fmap, rois, rel_inds[:, 1:], res.im_sizes)
        od, rd = model.predict(nf, ef, rel_inds, rois, res.im_sizes, _im_inds=res.im_inds.contiguous())
    return od.float().cpu().numpy(), rd.float().cpu().numpy(), rel_inds.cpu().numpy()


def bench_setup():
    """-> dict(model, batch, ref_od, ref_rd, ref_rel_inds, oracle_seconds): built once per process"""
    if 'setup' not in _STATE:
        import sgg_amd
        from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
        model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls'))
        sd = {k: v.clone() for k, v in model.state_dict().items()}
        model.to(DEV).eval()
        batch = synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=111)          # bench.py's rank-0 batch
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        t0 = time.time()
        with torch.no_grad():
            ref = O.forward_gtbox(batch[0], batch[3], batch[4], batch[5], sd, mode='sgcls')
        _STATE['setup'] = dict(model=model, batch=batch, ref_od=ref['rm_obj_dists'].numpy(), ref_rd=ref['rel_dists'].numpy(),
                               ref_rel_inds=ref['rel_inds'], oracle_seconds=round(time.time() - t0, 1))
    return _STATE['setup']


def mode_report(name):
    """name in f32 / x3 / f16 / bf16 -> the error figures of that mode's logits against the oracle (dumped to gpurun_out/)"""
    s = bench_setup()
    model = s['model']
    if name == 'x3':
        model.set_compute_dtype(torch.float32, split3=True)
    else:
        model.set_compute_dtype({'f32': torch.float32, 'f16': torch.float16, 'bf16': torch.bfloat16}[name])
    try:
        od, rd, rel_inds = logits(model, s['batch'])
    finally:
        model.set_compute_dtype(torch.float16)
    np.testing.assert_array_equal(rel_inds, s['ref_rel_inds'])
    assert od.shape == (256, 151) and rd.shape == (7936, 51)
    ref_od, ref_rd = s['ref_od'], s['ref_rd']
    e_o, e_r = np.abs(od - ref_od), np.abs(rd - ref_rd)
    rep = {'obj_max_abs': float(e_o.max()), 'obj_mean_abs': float(e_o.mean()), 'rel_max_abs': float(e_r.max()),
           'rel_mean_abs': float(e_r.mean()), 'obj_logit_absmax': float(np.abs(ref_od).max()),
           'rel_logit_absmax': float(np.abs(ref_rd).max()),
           'obj_argmax_agreement': float((od[:, 1:].argmax(1) == ref_od[:, 1:].argmax(1)).mean()),
           'rel_argmax_agreement': float((rd[:, 1:].argmax(1) == ref_rd[:, 1:].argmax(1)).mean())}
    dump('r06_parity_bench_config.json', {'config': '8 x 592x592, 32 boxes, 992 edges per image, seed 111', 'oracle_seconds': s['oracle_seconds'],
                                          name: rep}, merge=True)
    print(name, json.dumps(rep, indent=1))
    return rep
