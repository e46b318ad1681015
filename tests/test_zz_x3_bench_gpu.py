"""The x3 mode at the BENCHMARKED configuration.  Collected last on purpose ('zz'): the driver runs `pytest -x`, and in
round 4 this mode's assertion stopped the run in front of 90 tests.

 * test_x3_bench_config_is_bit_reproducible: two x3 forwards of the seed-111 bench batch in one process with the caching allocator
   poisoned in between (cache emptied, 10 GiB filled with a byte pattern and freed: every later torch.empty comes out of those blocks)
   -- logits equal bit for bit;
 * test_x3_bench_config_logits_within_1e3: against the fp32 CPU oracle, the north star's clause.
On failure the per-stage table of tools/x3_repro.py (every contraction's output against the exact-fp32 mode's) is written to
gpurun_out/x3_stage_diag.json, so that a red run names the first stage that went wrong."""
import json
import os

import pytest
import torch

from tests.parity_common import DEV, bench_setup, dump, logits, mode_report

pytestmark = pytest.mark.gpu


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


def _diag(tag):
    from tools import x3_repro
    s = bench_setup()
    rows, logit_err, _, _ = x3_repro.staged_compare(s['model'], s['batch'])
    s['model'].set_compute_dtype(torch.float16)
    dump('x3_stage_diag.json', {tag: dict(stages=rows, logits=logit_err)}, merge=True)
    x3_repro.print_rows(rows)
    print(json.dumps(logit_err))


def test_x3_bench_config_is_bit_reproducible():
    _gpu()
    from tools.x3_repro import poison
    s = bench_setup()
    model, batch = s['model'], s['batch']
    outs = []
    try:
        for byte in (None, 0x7f, 0x3c):
            if byte is not None:
                poison(byte)
            model.set_compute_dtype(torch.float32, split3=True)
            od, rd, _ = logits(model, batch)
            outs.append((od, rd))
    finally:
        model.set_compute_dtype(torch.float16)
    same = all((o[0] == outs[0][0]).all() and (o[1] == outs[0][1]).all() for o in outs[1:])
    if not same:
        _diag('bit_repro')
    assert same, 'two x3 forwards of the same batch differ'


@pytest.mark.parametrize('mode', ['x3', 'f16'])
def test_bench_config_forward_is_the_same_bits_over_many_runs(mode):
    """30 forwards per mode (two streams, as shipped), some of them started on a busy GPU: one distinct result.  (Before the round-5 fix of
    the gate kernels' dot products 5 - 10 % of the x3 forwards differed, by up to 0.03 in a logit.)"""
    _gpu()
    s = bench_setup()
    model, batch = s['model'], s['batch']
    busy = torch.randn(4096, 4096, device=DEV)
    try:
        if mode == 'x3':
            model.set_compute_dtype(torch.float32, split3=True)
        else:
            model.set_compute_dtype(torch.float16)
        first, differing = None, 0
        for k in range(30):
            if k % 4 == 2:
                for _ in range(6):
                    busy = (busy @ busy).clamp_(-1, 1)
            od, rd, _ = logits(model, batch)
            if first is None:
                first = (od, rd)
            elif not ((od == first[0]).all() and (rd == first[1]).all()):
                differing += 1
    finally:
        model.set_compute_dtype(torch.float16)
    assert differing == 0, '%d of 29 forwards differ from the first' % differing


def test_x3_bench_config_logits_within_1e3():
    _gpu()
    x3 = mode_report('x3')
    ok = x3['obj_max_abs'] <= 1e-3 and x3['rel_max_abs'] <= 1e-3
    if not ok:
        _diag('within_1e3')
    assert ok, x3                                       # the north star's bar, in the fast mode as well
    assert x3['obj_argmax_agreement'] == 1.0 and x3['rel_argmax_agreement'] == 1.0, x3


def test_x3_bench_config_head_gradients_match_exact_fp32():
    """the x3 backward splits dY into f16 halves; unscaled head gradients at this size are 1e-6 .. 1e-8 per element (no hi half
    below 6e-8).  With the Trainer's loss scale (now applied in the x3 mode as in f16) every head gradient agrees with the exact-fp32 mode's
    per tensor: max |g_x3 - g_f32| <= 5e-3 max |g_f32| (measured on MI355X: 2e-5 .. 5e-4 for the Linear / GRU / gate tensors, 2.3e-3 for the
    node branch's fc7), 3e-2 for the rect convolution's layers (measured 2e-3 .. 1.2e-2: their gradients pass through two batch-statistic
    BatchNorms, whose backward subtracts two nearly equal means, and ReLUs whose pre-activations near zero may fall on the other side in the two
    modes).  An operand half lost to f16's range -- what this test is for -- shows as an error of order 1."""
    _gpu()
    from sgg_amd.trainer import Trainer
    s = bench_setup()
    model, batch = s['model'], s['batch']
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    drop = model.dropout_p
    grads = {}
    try:
        model.train()
        model.dropout_p = 0.0
        for split in (False, True):
            model.load_state_dict(sd)
            model.set_compute_dtype(torch.float32, split3=split)
            tr = Trainer(model, lr=1e-3)
            assert tr.loss_scale == (tr.loss_scale_f16 if split else 1.0)
            res = model([tuple(batch)])
            loss = tr.losses(res)
            model.zero_grad()
            (loss * tr.loss_scale).backward()
            grads[split] = {n: p.grad.detach().clone() / tr.loss_scale for n, p in model.named_parameters() if p.grad is not None}
            del tr
    finally:
        model.load_state_dict(sd)
        model.eval()
        model.dropout_p = drop
        model.zero_grad()
        model.set_compute_dtype(torch.float16)
    assert len(grads[True]) == len(grads[False]) >= 30
    worst = {}
    for n, ge in grads[False].items():
        scale = float(ge.abs().max())
        worst[n] = float((grads[True][n] - ge).abs().max()) / scale if scale > 0 else 0.0
    dump('r06_x3_gradients_bench_config.json', {'relative_to_each_tensors_largest_gradient': worst})
    bound = lambda n: 3e-2 if n.startswith('union_boxes.conv.') else 5e-3
    bad = {n: v for n, v in worst.items() if v > bound(n)}
    assert not bad, bad
