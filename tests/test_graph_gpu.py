"""The train step replayed as hipGraphs (sgg_amd/graph_step.py) against the launch-by-launch step of Trainer:
the same kernels with the same arguments in the same stream order, so after n steps -- over two batch signatures, through the warm-up,
the captures and the replays -- every parameter, momentum buffer and BatchNorm statistic is bit-equal, and so is every step's loss."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
S = 160


def _gpu():
    if not torch.cuda.is_available():
        pytest.skip('no GPU')


def _batches(dev_images=True):
    from sgg_amd.rel_model_base import to_device_with_mirror
    from sgg_amd.synthetic import synthetic_batch
    out = []
    for k, counts in enumerate(((7, 7, 7), (7, 7, 7), (6, 8, 7), (7, 7, 7))):
        b = list(synthetic_batch(B=3, S=S, n_boxes=7, n_fg=3, seed=50 + k, counts=counts))
        b[0] = [(im * 255).round().to(torch.uint8).permute(1, 2, 0).contiguous().to(DEV) for im in b[0]]     # decoded u8 images, as the boundary hands them over
        b[3] = b[3].to(DEV)
        b[4], b[5] = to_device_with_mirror(b[4], DEV), to_device_with_mirror(b[5], DEV)
        out.append(tuple(b))
    return out


def _run(graph, steps, dtype, sd, batches, lr_drop_at=None):
    import sgg_amd
    from sgg_amd.synthetic import SyntheticData, init_weights
    from sgg_amd.trainer import Trainer
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls', min_size=S, max_size=S))
    if sd is not None:
        model.load_state_dict(sd)
    model.to(DEV)
    model.set_compute_dtype(dtype)
    tr = Trainer(model, lr=2e-2, pipeline=True, graph=graph)
    assert (tr.graphs is not None) == graph
    torch.manual_seed(1234)                    # the dropout seeds of both runs come from torch's CPU generator
    losses = []
    for i in range(steps):
        if lr_drop_at is not None and i == lr_drop_at:
            for g in tr.opt.param_groups:      # what a MultiStepLR milestone does: the graphs that baked the old rate must not be replayed
                g['lr'] *= 0.1
        losses.append(tr.step(batches[i % len(batches)]))
    tr.flush()
    torch.cuda.synchronize()
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    mom = {n: tr.opt.state[p]['momentum_buffer'].clone() for n, p in model.named_parameters() if p in tr.opt.state and 'momentum_buffer' in tr.opt.state[p]}
    stats = dict(tr.graphs.stats) if tr.graphs is not None else None
    disabled = tr.graphs.disabled if tr.graphs is not None else None
    return [float(x) for x in losses], state, mom, stats, disabled, tr.opt.steps


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16])
def test_graph_steps_equal_plain_steps_bit_for_bit(dtype):
    _gpu()
    os.environ['SGG_GRAPH_STRICT'] = '1'
    try:
        batches = _batches()
        steps = 26          # 4 batches in rotation, two signatures: both are warmed (3 plain steps each), captured (B, then A per predecessor) and replayed
        l0, s0, m0, _, _, n0 = _run(False, steps, dtype, None, batches, lr_drop_at=19)
        l1, s1, m1, stats, disabled, n1 = _run(True, steps, dtype, None, batches, lr_drop_at=19)
    finally:
        os.environ.pop('SGG_GRAPH_STRICT', None)
    assert disabled is None, disabled
    assert stats['replayed'] >= 12 and stats['captures'] >= 4, stats
    assert n0 == n1 == steps
    assert l0 == l1, [(i, a, b) for i, (a, b) in enumerate(zip(l0, l1)) if a != b][:5]
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k
    assert set(m0) == set(m1)
    for k in m0:
        assert torch.equal(m0[k], m1[k]), k
    assert all(torch.isfinite(v.float()).all() for v in s1.values())
    assert l1[-1] < l1[0]


def test_300_replayed_steps_without_any_device_synchronisation_equal_plain_steps():
    """300 steps (four batches, two signatures) with the periodic device-wide synchronisation OFF (SGG_GRAPH_SYNC_EVERY=0)
    and the host free to run 64 steps ahead: every loss, parameter and momentum buffer bit-equal to the launch-by-launch run."""
    _gpu()
    os.environ['SGG_GRAPH_STRICT'] = '1'
    os.environ['SGG_GRAPH_SYNC_EVERY'] = '0'
    os.environ['SGG_GRAPH_DEPTH'] = '64'
    try:
        batches = _batches()
        l0, s0, m0, _, _, n0 = _run(False, 300, torch.float16, None, batches)
        l1, s1, m1, stats, disabled, n1 = _run(True, 300, torch.float16, None, batches)
    finally:
        for k in ('SGG_GRAPH_STRICT', 'SGG_GRAPH_SYNC_EVERY', 'SGG_GRAPH_DEPTH'):
            os.environ.pop(k, None)
    assert disabled is None, disabled
    assert stats['replayed'] >= 280, stats
    assert n0 == n1 == 300
    assert l0 == l1, [(i, a, b) for i, (a, b) in enumerate(zip(l0, l1)) if a != b][:5]
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k
    for k in m0:
        assert torch.equal(m0[k], m1[k]), k


def test_graph_steps_with_the_forward_in_two_graphs_equal_plain_steps():
    """SGG_GRAPH_VSPLIT=4: graph V cut after conv2_2, the update released after the first part -- scheduling only, the same bits"""
    _gpu()
    os.environ['SGG_GRAPH_STRICT'] = '1'
    os.environ['SGG_GRAPH_VSPLIT'] = '4'
    try:
        batches = _batches()
        l0, s0, m0, _, _, _ = _run(False, 14, torch.float16, None, batches)
        l1, s1, m1, stats, disabled, _ = _run(True, 14, torch.float16, None, batches)
    finally:
        os.environ.pop('SGG_GRAPH_STRICT', None)
        os.environ.pop('SGG_GRAPH_VSPLIT', None)
    assert disabled is None, disabled
    assert stats['replayed'] >= 4, stats
    assert l0 == l1
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k
    for k in m0:
        assert torch.equal(m0[k], m1[k]), k


def test_graph_step_falls_back_on_batches_it_does_not_take():
    """host-resident images, and a relation list with two relations on one ordered pair: plain steps, same results as a trainer without graphs"""
    _gpu()
    from sgg_amd.rel_model_base import to_device_with_mirror
    batches = _batches()
    odd = list(batches[0])
    rels = odd[5]._sgg_host.clone()
    rels = torch.cat((rels, rels[:1]), 0)                 # a duplicate relation: one more label row, no cached tables
    rels[-1, 3] = (rels[-1, 3] % 50) + 1
    order = torch.argsort(rels[:, 0], stable=True)
    odd[5] = to_device_with_mirror(rels[order].contiguous(), DEV)
    host = list(batches[1])
    host[0] = [im.cpu() for im in host[0]]
    seq = [batches[0]] * 5 + [tuple(odd), batches[0], tuple(host), batches[0], batches[0]]
    l0, s0, _, _, _, _ = _run(False, len(seq), torch.float16, None, seq)
    l1, s1, _, stats, disabled, _ = _run(True, len(seq), torch.float16, None, seq)
    assert disabled is None, disabled
    assert stats['replayed'] >= 4 and stats['plain'] >= 5, stats
    assert l0 == l1
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


def _other_signatures(n):
    """n batches with n distinct boxes-per-image signatures (none of them _batches()'s): each makes one entry in the model's candidate-edge
    cache, its segment cache and the pair tables"""
    from sgg_amd.rel_model_base import to_device_with_mirror
    from sgg_amd.synthetic import synthetic_batch
    out = []
    for k in range(n):
        counts = (3 + k % 6, 3 + (k // 6) % 6, 9 + k // 36)
        b = list(synthetic_batch(B=3, S=S, n_boxes=7, n_fg=2, seed=900 + k, counts=counts))
        b[0] = [(im * 255).round().to(torch.uint8).permute(1, 2, 0).contiguous().to(DEV) for im in b[0]]
        b[3] = b[3].to(DEV)
        b[4], b[5] = to_device_with_mirror(b[4], DEV), to_device_with_mirror(b[5], DEV)
        out.append(tuple(b))
    assert len({tuple(torch.bincount(b[4]._sgg_host[:, 0]).tolist()) for b in out}) == n
    return out


def test_replays_survive_the_eviction_of_the_per_signature_caches():
    """a captured step reads the candidate-edge list, its CSR lists and the pair tables at the addresses they had at capture.
    Those caches evict by clearing themselves (> 32 / > 64 signatures).  Capture one signature, push 70 OTHER signatures through the plain
    path (every cache is cleared at least once, the freed index tensors are overwritten with garbage by allocations of the same sizes), then
    replay the first: bit-equal to a trainer without graphs fed the same sequence."""
    _gpu()
    import gc
    os.environ['SGG_GRAPH_STRICT'] = '1'
    try:
        first = _batches()[0]
        others = _other_signatures(70)
        seq = [first] * 6 + others + [first] * 4
        l0, s0, m0, _, _, _ = _run(False, len(seq), torch.float16, None, seq)
        # (the run with graphs: after the 70 plain steps, scribble over whatever the allocator holds free -- a use-after-free then reads garbage
        # indices: wrong losses or a fault, not a silent pass)
        import sgg_amd
        from sgg_amd import pairing
        from sgg_amd.synthetic import SyntheticData, init_weights
        from sgg_amd.trainer import Trainer
        model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls', min_size=S, max_size=S)).to(DEV)
        model.set_compute_dtype(torch.float16)
        tr = Trainer(model, lr=2e-2, pipeline=True, graph=True)
        torch.manual_seed(1234)
        l1 = []
        for i, b in enumerate(seq):
            if i == 6 + len(others):
                assert tr.graphs.stats['replayed'] >= 2, tr.graphs.stats
                key0 = [k for k in tr.graphs.B][0]
                assert not any(k[0] == tuple((i_, 7) for i_ in range(3)) for k in model._graph_cache), 'the cache was not evicted: the test tests nothing'
                gc.collect()
                torch.cuda.synchronize()
                junk = [torch.full((n,), 0x7fffffff, dtype=torch.int32, device=DEV) for n in (64, 256, 1024, 4096, 16384, 65536) for _ in range(64)]
                del junk
                torch.cuda.synchronize()
            l1.append(float(tr.step(b)))
        tr.flush()
        torch.cuda.synchronize()
        stats, disabled = dict(tr.graphs.stats), tr.graphs.disabled
        s1 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    finally:
        os.environ.pop('SGG_GRAPH_STRICT', None)
    assert disabled is None, disabled
    assert stats['replayed'] >= 6, stats
    assert l0 == l1, [(i, a, b) for i, (a, b) in enumerate(zip(l0, l1)) if a != b][:5]
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


def test_flush_after_a_scheduler_step_applies_the_pending_update_with_the_old_rate():
    """scheduler.step() between the last step() of an epoch and flush() must not change that step's update (the plain pipelined
    path queued it with the old rate at the end of the step)"""
    _gpu()
    import sgg_amd
    from sgg_amd.synthetic import SyntheticData, init_weights
    from sgg_amd.trainer import Trainer
    batches = _batches()
    out = []
    for graph in (False, True):
        model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls', min_size=S, max_size=S)).to(DEV)
        model.set_compute_dtype(torch.float16)
        tr = Trainer(model, lr=2e-2, pipeline=True, graph=graph)
        torch.manual_seed(77)
        for i in range(8):
            tr.step(batches[0])
        if graph:
            assert tr.graphs.stats['replayed'] >= 3 and tr.graphs.pending is not None
        for g in tr.opt.param_groups:
            g['lr'] *= 0.1                  # the epoch's scheduler step
        tr.flush()
        torch.cuda.synchronize()
        out.append({k: v.detach().clone() for k, v in model.state_dict().items()})
    for k in out[0]:
        assert torch.equal(out[0][k], out[1][k]), k


@pytest.mark.parametrize('mode', ['sgcls', 'predcls'])
def test_eval_forward_graph_equals_plain_forward_bit_for_bit(mode):
    """sgg_amd/graph_forward.py: model([batch]) in eval mode with device-resident inputs -- two plain calls per signature, then one replayed
    hipGraph per call; the five result arrays equal the plain forward's (SGG_GRAPH=0) bit for bit, also after the weights changed"""
    _gpu()
    import numpy as np
    import sgg_amd
    from sgg_amd.synthetic import SyntheticData, init_weights
    batches = _batches()
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode=mode, min_size=S, max_size=S)).to(DEV).eval()
    model.set_compute_dtype(torch.float16)
    seq = [batches[0], batches[1], batches[0], batches[2], batches[1], batches[2], batches[3], batches[2], batches[0], batches[2]]

    def run_all(graph):
        os.environ['SGG_GRAPH'] = '1' if graph else '0'
        os.environ['SGG_GRAPH_STRICT'] = '1'
        try:
            model.enable_eval_graphs(graph)
            outs = []
            with torch.no_grad():
                for b in seq:
                    outs.append(model([b]))
            eg = model.__dict__.get('_eval_graphs')
            return outs, (dict(eg.stats) if hasattr(eg, 'stats') else None), (eg.disabled if hasattr(eg, 'stats') else None)
        finally:
            os.environ.pop('SGG_GRAPH', None)
            os.environ.pop('SGG_GRAPH_STRICT', None)
    plain, _, _ = run_all(False)
    graphed, stats, disabled = run_all(True)
    assert disabled is None, disabled
    assert stats['replayed'] >= 5 and stats['captures'] == 2, stats
    for a, b in zip(plain, graphed):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
    # the weights move (what a training step between two evaluations does): the graphs are made again and follow
    with torch.no_grad():
        model.rel_fc.weight.mul_(1.5)
    plain2, _, _ = run_all(False)
    graphed2, stats2, _ = run_all(True)
    assert stats2['replayed'] >= 5
    assert not all(np.array_equal(x, y) for x, y in zip(plain[0], plain2[0]))
    for a, b in zip(plain2, graphed2):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
