"""CPU, world_size 2 over gloo: image sharding, max-over-ranks timing, loss-normaliser and gradient all-reduce."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from sgg_amd import dist as D
    r, _, w = D.init('gloo')
    lo, hi = D.shard_range(13, r, w)
    mx = D.max_over_ranks(1.0 + r)
    tot = D.sum_over_ranks([hi - lo, 2.0 * r])
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.Linear(16, 3))
    x = torch.full((3, 8), float(r + 1))
    lin(x).sum().backward()
    local = [p.grad.clone() for p in lin.parameters()]
    D.GradBuckets(lin.parameters(), big_bytes=256).all_reduce(average=True)   # Linear(8,16).weight is 'big'
    avg = [p.grad.clone() for p in lin.parameters()]
    # the DP train-step form: SUM in a different wire dtype, big tensors handed back as buffers (no copy into .grad)
    for p, g in zip(lin.parameters(), local):
        p.grad = g.clone()
    gb = D.GradBuckets(lin.parameters(), big_bytes=256, comm_dtype=torch.float64)
    first = next(iter(lin.parameters()))
    gb.start(first, first.grad)                      # early launch from inside the backward (overlap path)
    direct = gb.all_reduce(average=False)
    summed = [(direct[p] if p in direct else p.grad).float() for p in lin.parameters()]
    ok = len(direct) == 4 and all(torch.allclose(s, a * 2, atol=1e-6) for s, a in zip(summed, avg))
    ok = ok and all(t.data_ptr() % 16 == 0 for t in direct.values())     # bucket views stay 16-byte aligned (3-element bias)
    # sharded form: the big tensor (Linear(8,16).weight: 128 elements) is reduce-scattered -- each rank gets the summed gradient
    # of ITS half only, early-started or not; the small ones still arrive whole; a tensor that does not split evenly is all-reduced
    for early in (True, False):
        for p, g in zip(lin.parameters(), local):
            p.grad = g.clone()
        gs = D.GradBuckets(lin.parameters(), big_bytes=256, shard=True)
        ok = ok and gs.shard_of(first) == (64 * r, 64 * (r + 1)) and all(gs.shard_of(p) is None for p in list(lin.parameters())[1:])
        if early:
            gs.start(first, first.grad)
        part = gs.all_reduce(average=False)
        ok = ok and part[first].shape == (64,) and torch.allclose(part[first], (avg[0] * 2).reshape(-1)[64 * r:64 * (r + 1)], atol=1e-6)
        ok = ok and all(torch.allclose(part[p].float(), a * 2, atol=1e-6) for p, a in list(zip(lin.parameters(), avg))[1:])
    odd = torch.nn.Parameter(torch.ones(100))                              # 100 elements do not split into 2 x (multiple of 8)
    odd.grad = torch.full((100,), float(r + 1))
    go = D.GradBuckets([odd], big_bytes=256, shard=True)
    ok = ok and go.shard_of(odd) is None and torch.allclose(go.all_reduce(average=False).get(odd, odd.grad), torch.full((100,), 3.0))
    try:
        gs.all_reduce(average=True)
        ok = False
    except ValueError:
        pass
    q.put((r, (lo, hi), mx, tot, [g.tolist() for g in local], [g.tolist() for g in avg], ok))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=120) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, m0, t0, l0, g0, ok0), (r1, s1, m1, t1, l1, g1, ok1) = out
    assert ok0 and ok1
    assert s0 == (0, 7) and s1 == (7, 13)                 # contiguous, covers all images once
    assert m0 == m1 == 2.0                                # max over ranks
    assert t0 == t1 == [13.0, 2.0]
    for a, b, ga, gb in zip(l0, l1, g0, g1):
        exp = (torch.tensor(a) + torch.tensor(b)) / 2
        torch.testing.assert_close(torch.tensor(ga), exp)
        torch.testing.assert_close(torch.tensor(gb), exp)


def test_shard_range_properties():
    from sgg_amd.dist import shard_range
    for n in (0, 1, 7, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


# ------------------------------------------------------------------ the Trainer's data-parallel logic at world size 2
class _Heads(torch.nn.Module):
    """Stand-in for the model on the CPU: two linear heads over given features, returning what Trainer.losses reads."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(5)
        self.obj_fc = torch.nn.Linear(6, 7)
        self.rel_fc = torch.nn.Linear(6, 5)

    def forward(self, xo, xr, obj_labels, rel_labels):
        from sgg_amd.result import Result
        return Result(rm_obj_dists=self.obj_fc(xo), rel_dists=self.rel_fc(xr), rm_obj_labels=obj_labels, rel_labels=rel_labels)


def _split_batch():
    """A global batch and a RAGGED split of it over two ranks: different object / edge / FG counts per rank; rank 1 of the
    second case has no FG edge at all (the M_FG = 0 branch of lib/losses.py:50)."""
    g = torch.Generator().manual_seed(9)
    xo, xr = torch.randn(11, 6, generator=g), torch.randn(23, 6, generator=g)
    ol = torch.randint(0, 7, (11,), generator=g)
    pred = torch.randint(1, 5, (23,), generator=g)
    cases = []
    for n0, m0, fg in ((4, 9, torch.rand(23, generator=g) < 0.3), (7, 15, torch.arange(23) < 6)):
        p = torch.where(fg, pred, torch.zeros_like(pred))
        rl = torch.stack((torch.zeros(23, dtype=torch.long), torch.arange(23), torch.arange(23), p), 1)
        cases.append((xo, xr, ol, rl, n0, m0))
    return cases


def _trainer_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from sgg_amd import dist as D
    from sgg_amd.trainer import Trainer
    D.init('gloo')
    out = {}
    for ci, (xo, xr, ol, rl, n0, m0) in enumerate(_split_batch()):
        so, sr = (slice(0, n0), slice(0, m0)) if rank == 0 else (slice(n0, None), slice(m0, None))
        for lt, lw in (('baseline', (1, 1, 0.5)), ('dnorm', (1.0, 0.7, 2.0)), ('dnorm-fgbg', (0.3, 1.5, 1.0))):
            model = _Heads()
            tr = Trainer(model, loss_type=lt, loss_weights=lw, comm_dtype=None)
            assert tr.dist_on and tr.world == 2
            loss = tr.losses(model(xo[so], xr[sr], ol[so], rl[sr]))
            loss.backward()
            grads = tr.buckets.all_reduce(average=False)            # SUM, as Trainer.step issues it
            out[(ci, lt)] = (float(loss.detach()), {n: grads[p].tolist() for n, p in model.named_parameters()})
    # a pass that ONE rank runs on its own (bench.py's profiling pass) must not issue a collective: rank 0 computes a loss
    # inside local_only() while rank 1 goes straight to the barrier -- a stray all-reduce would pair with it and corrupt / hang
    model = _Heads()
    tr = Trainer(model, loss_type='baseline', comm_dtype=None)
    xo, xr, ol, rl, n0, m0 = _split_batch()[0]
    if rank == 0:
        with tr.local_only():
            assert not tr.dist_on and tr.world == 1 and model._grad_ready_hook is None and model._bn_sync is None
            solo = float(tr.losses(model(xo[:n0], xr[:m0], ol[:n0], rl[:m0])).detach())
        assert tr.dist_on and tr.world == 2 and model._grad_ready_hook is not None
        out['solo'] = solo
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)                                              # the next collective both ranks issue must still pair up
    out['after'] = float(t)
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_losses_global_normalisers_world_size_2():
    """lib/losses.py:34-63,74 on a split batch: per-rank losses with GLOBAL normalisers + all-reduce(SUM) of the gradients
    = the single-process loss / gradient on the concatenated batch, for all three loss types (oracle restatement, pinned by
    tests/golden/losses.npz)."""
    from oracle import sgg_oracle as O
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0]['after'] == res[1]['after'] == 3.0
    for ci, (xo, xr, ol, rl, n0, m0) in enumerate(_split_batch()):
        for lt, lw in (('baseline', (1, 1, 0.5)), ('dnorm', (1.0, 0.7, 2.0)), ('dnorm-fgbg', (0.3, 1.5, 1.0))):
            model = _Heads()
            r = model(xo, xr, ol, rl)
            ref = O.node_losses(r.rm_obj_dists, r.rm_obj_labels) + O.edge_losses(r.rel_dists, r.rel_labels[:, -1], lt, lw)
            ref.backward()
            l0, g0 = res[0][(ci, lt)]
            l1, g1 = res[1][(ci, lt)]
            assert abs(l0 + l1 - float(ref)) < 1e-5 * max(1.0, abs(float(ref))), (ci, lt, l0, l1, float(ref))
            for n, p in model.named_parameters():
                torch.testing.assert_close(torch.tensor(g0[n]), p.grad, atol=1e-6, rtol=1e-5)
                torch.testing.assert_close(torch.tensor(g1[n]), p.grad, atol=1e-6, rtol=1e-5)
    # the solo loss uses rank 0's LOCAL normalisers
    xo, xr, ol, rl, n0, m0 = _split_batch()[0]
    model = _Heads()
    r = model(xo[:n0], xr[:m0], ol[:n0], rl[:m0])
    ref = O.node_losses(r.rm_obj_dists, r.rm_obj_labels) + O.edge_losses(r.rel_dists, r.rel_labels[:, -1], 'baseline', (1, 1, 1))
    assert abs(res[0]['solo'] - float(ref)) < 1e-5


def test_bench_launcher_spawns_ranks_dry():
    """`python bench.py --gpus 2 --dry` (no rank environment): the parent starts 2 ranks before touching the GPU, they form a
    group, and rank 0's JSON line -- the last line on stdout -- says n_gpus 2.  A mismatch between --gpus and WORLD_SIZE fails."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--dry', '--steps', '4'], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['steps'] == 4 and line['config']['global_batch'] == 16
    bad = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--dry'], env=dict(env, WORLD_SIZE='2', RANK='0'),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and 'WORLD_SIZE' in bad.stderr
    # the contract of torch.distributed.run (what the driver uses for N > 1) gives the same line
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--dry'], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([l for l in r.stdout.strip().splitlines() if l.startswith('{')][-1])['n_gpus'] == 2


def test_bench_one_gpu_run_is_guarded_and_falls_back_once():
    """`python bench.py` on one GPU runs in a child process started before anything touches the GPU; a child that dies (here: no GPU at
    all) is repeated ONCE with the train step launched kernel by kernel (SGG_GRAPH=0) and its exit code is passed on."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'SGG_BENCH_CHILD',
                                                            'SGG_GRAPH', 'SGG_BENCH_GUARD')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '1', '--warmup', '0', '--no-f32', '--no-cpu-baseline'],
                       env=env, capture_output=True, text=True, timeout=600)
    if torch.cuda.is_available():
        pytest.skip('a GPU is here: the run succeeds (tests/-m gpu cover it)')
    assert r.returncode != 0
    assert r.stderr.count('once more with the train step launch by launch') == 1, r.stderr[-1500:]


def test_bf16_wire_sum_error_bound_8_ranks():
    """GradBuckets sums the big gradients in bf16 on the wire.  Bound for 8 ranks, worst case = a chain of 7 bf16 additions (ring
    reduce-scatter: every hop adds two bf16 numbers and rounds): against the fp32 sum of the fp32 gradients the error stays
    below 8 half-ulps of the LARGEST partial sum per element, and below 1 % of the gradient's norm overall."""
    g = torch.Generator().manual_seed(3)
    base = torch.randn(1 << 16, generator=g)                               # the common part of the ranks' gradients
    grads = [base + 0.5 * torch.randn(1 << 16, generator=g) for _ in range(8)]
    exact = torch.stack(grads).double().sum(0)
    acc = grads[0].to(torch.bfloat16)
    part = grads[0].double().abs()
    for t in grads[1:]:
        acc = acc + t.to(torch.bfloat16)                                   # bf16 + bf16 -> bf16 (one rounding per hop)
        part = torch.maximum(part, acc.double().abs())
    err = (acc.double() - exact).abs()
    eps = 2.0 ** -8                                                        # bf16: 8 significand bits -> half-ulp = 2^-9 relative
    assert bool((err <= 8 * eps * part + 1e-30).all())
    assert float(err.norm() / exact.norm()) < 1e-2


def _shard8_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from sgg_amd import dist as D
    D.init('gloo')
    g = torch.Generator().manual_seed(7)                       # the same parameters on every rank, rank-dependent gradients
    shapes = [(64, 40), (24, 8), (13, 9), (640,), (100,), (3,), (8, 8, 3, 3)]    # 2560, 192, 117, 640, 100, 3, 576 elements
    params = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    grads = [torch.randn(s, generator=g) for s in shapes]
    for p, gr in zip(params, grads):
        p.grad = gr * (rank + 1)
    total = sum(range(1, world + 1))
    gb = D.GradBuckets(params, big_bytes=400, shard=True)        # 'big' = >= 100 fp32 elements
    ok = [p.numel() >= 100 for p in params] == [gb.is_big(p) for p in params]
    want = {0: True, 1: True, 2: False, 3: True, 4: False, 6: True}   # sharded iff the element count splits into 8 x (multiple of 8)
    for i, sharded in want.items():
        rng = gb.shard_of(params[i])
        n = params[i].numel()
        ok = ok and ((rng == (rank * n // world, (rank + 1) * n // world)) if sharded else rng is None)
    gb.start(params[0], params[0].grad)                          # one of them early, from "inside the backward"
    out = gb.all_reduce(average=False)
    for i, (p, gr) in enumerate(zip(params, grads)):
        exp = (gr * total).reshape(-1)
        rng = gb.shard_of(p)
        got = out[p] if p in out else p.grad
        if rng is not None:
            ok = ok and got.shape == (rng[1] - rng[0],) and torch.allclose(got.float(), exp[rng[0]:rng[1]], atol=1e-5)
        else:
            ok = ok and torch.allclose(got.float().reshape(-1), exp, atol=1e-5)
        ok = ok and got.data_ptr() % 16 == 0
    # what FusedSGD does with the parts afterwards: update its part, all-gather the whole tensor back in place
    for p in (params[0], params[3], params[6]):
        lo, hi = gb.shard_of(p)
        flat = p.data.view(-1)
        flat[lo:hi] -= 0.1 * out[p].float()
        dist.all_gather_into_tensor(flat, flat[lo:hi].clone())
    q.put((rank, bool(ok), [p.detach().reshape(-1).tolist() for p in params]))      # (plain lists: no shared-memory handles to outlive the rank)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_grad_buckets_world_size_8():
    """GradBuckets(shard=True) on EIGHT gloo ranks (BASELINE configs[3]: 8 x MI355X): which tensors split (element counts that are 8 x a
    multiple of 8) and which fall back to the all-reduce (117, 100 elements), every rank's part of the summed gradient, early-started or
    not, 16-byte aligned views, and the update + in-place all-gather round trip leaving the same parameters on every rank."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard8_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r for r, _, _ in out] == list(range(8)) and all(ok for _, ok, _ in out)
    for _, _, ps in out[1:]:
        assert ps == out[0][2]


def test_trainer_losses_never_read_device_values_on_the_host():
    """no `.item()` / `.tolist()` / float(tensor) between forward and backward -- the normalisers of all three loss
    forms stay device tensors (the world-size-2 test above checks their values, incl. M_FG = 0 on one rank)."""
    import inspect
    import re
    from sgg_amd.trainer import Trainer
    for fn in (Trainer.losses, Trainer._fused_losses):
        src = re.sub(r'""".*?"""', '', inspect.getsource(fn), flags=re.S)
        src = '\n'.join(l.split('#')[0] for l in src.splitlines())
        for needle in ('.item()', '.tolist()', '.cpu()', 'sum_over_ranks', 'synchronize'):
            assert needle not in src, (fn.__name__, needle)
