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
