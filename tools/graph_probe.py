"""How fast do the two hipGraphs of a train step (sgg_amd/graph_step.py) replay?  One batch signature, graphs primed, then: whole steps,
graphs U (update), V (VGG forward), B (head forward + backward) alone -- each N times back to back, per-launch time.
    python tools/graph_probe.py [steps]          (environment: SGG_GRAPH=0 for the launch-by-launch step; HIP's DEBUG_* graph switches)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import sgg_amd
from sgg_amd.rel_model_base import to_device_with_mirror
from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
from sgg_amd.trainer import Trainer


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    dev = 'cuda:0'
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls')).to(dev).eval()
    model.set_compute_dtype(torch.float16)
    b = list(synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=111))
    b[0] = [(im * 255).round().to(torch.uint8).permute(1, 2, 0).contiguous().to(dev) for im in b[0]]
    b[3], b[4], b[5] = b[3].to(dev), to_device_with_mirror(b[4], dev), to_device_with_mirror(b[5], dev)
    b = tuple(b)
    rot = [b]
    if os.environ.get('ROTATE'):            # bench.py's rotation: three batches of one signature, a fourth with 30..34 boxes per image
        rot = []
        for k in range(4):
            hb = list(synthetic_batch(B=8, S=592, n_boxes=32, n_fg=6, seed=111 + 1000 * k, counts=(30, 34, 32, 32, 31, 33, 32, 32) if k == 3 else None))
            hb[0] = [(im * 255).round().to(torch.uint8).permute(1, 2, 0).contiguous().to(dev) for im in hb[0]]
            hb[3], hb[4], hb[5] = hb[3].to(dev), to_device_with_mirror(hb[4], dev), to_device_with_mirror(hb[5], dev)
            rot.append(tuple(hb))
    tr = Trainer(model, lr=1e-3, pipeline=True)
    for k in range(10 if len(rot) == 1 else 40):
        tr.step(rot[k % len(rot)])
    torch.cuda.synchronize()
    if len(rot) > 1:
        per = []
        for k in range(24):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tr.step(rot[k % len(rot)])
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            per.append('%d:%.1f/%.1f' % (k % len(rot), (t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
        print('rotation, synchronised per step (batch: issue / total ms):', ' '.join(per))
        kk = [0]

        def one():
            tr.step(rot[kk[0] % len(rot)])
            kk[0] += 1
        i, t = timed(one, steps)
        print('rotation: issue %.3f ms, %.3f ms per step (%s)' % (i, t, tr.graphs.stats if tr.graphs is not None else 'launch by launch'))
        if os.environ.get('STAGER'):        # the bench's host-resident feed: DeviceStager.prefetch on a worker thread
            from sgg_amd.blob import DeviceStager
            host = []
            for hb in rot:
                t_ = list(hb)
                t_[0] = [im.cpu() for im in hb[0]]
                t_[3], t_[4], t_[5] = hb[3].cpu(), hb[4]._sgg_host, hb[5]._sgg_host
                host.append(tuple(t_))
            _shift = [torch.cuda.Stream() for _ in range(int(os.environ.get('STAGER_SHIFT', '0')))]      # (which pool stream -- which hardware queue -- the H2D copies get)
            stager = DeviceStager(dev, slots=16)
            n = 3 * steps
            print('stager feed ...', flush=True)
            t0 = time.perf_counter()
            for k, b_ in enumerate(stager.prefetch((host[i % 4] for i in range(n)))):
                tr.step(b_)
                if k % 20 == 0:
                    torch.cuda.synchronize()
                    print('  step %d ok' % k, flush=True)
            torch.cuda.synchronize()
            print('stager feed: %.3f ms per step' % ((time.perf_counter() - t0) / n * 1e3))

    if os.environ.get('FLUSH_PROBE'):
        import gc
        tr.flush()
        torch.cuda.synchronize()
        print('flush ok', flush=True)
        for k in range(5):
            tr.step(rot[k % len(rot)])
            if os.environ.get('FLUSH_PROBE') == '2':
                torch.cuda.synchronize()
                print('  step %d after the flush ok' % k, flush=True)
        torch.cuda.synchronize()
        print('5 steps after the flush ok', flush=True)
        gc.collect()
        gc.freeze()
        torch.cuda.synchronize()
        for k in range(8):
            tr.step(rot[k % len(rot)])
        torch.cuda.synchronize()
        print('8 more steps after collect + freeze ok', flush=True)
    if os.environ.get('GC_PROBE'):
        # what would a full collection free now?  (bench.py collects + freezes after its warm-up; round 5: replays faulted afterwards)
        import gc
        gc.set_debug(gc.DEBUG_SAVEALL)
        n = gc.collect()
        gc.set_debug(0)
        seen = {}
        for o in gc.garbage:
            if torch.is_tensor(o) and o.is_cuda:
                k = (tuple(o.shape), str(o.dtype))
                seen[k] = seen.get(k, 0) + 1
        kinds = {}
        for o in gc.garbage:
            kinds[type(o).__name__] = kinds.get(type(o).__name__, 0) + 1
        print('gc.collect(): %d unreachable objects; GPU tensors among them: %s' % (n, sorted(seen.items(), key=lambda kv: -kv[1])[:30]))
        print('types:', sorted(kinds.items(), key=lambda kv: -kv[1])[:25])
        gc.garbage.clear()
        gc.collect()
        for k in range(8):
            tr.step(rot[k % len(rot)])
        torch.cuda.synchronize()
        print('8 steps after the collection: ok')
    i, t = timed(lambda: tr.step(b), steps)
    print('whole step: issue %.3f ms, %.3f ms per step (%s)' % (i, t, tr.graphs.stats if tr.graphs is not None else 'launch by launch'))
    if tr.graphs is not None and tr.graphs.B:
        g = tr.graphs
        bb = next(iter(g.B.values()))
        i, t = timed(lambda: g._launch_B(bb), steps)
        print('graph B alone (head forward + loss + backward): issue %.3f ms, %.3f ms per replay' % (i, t))
        for k, a in g.U.items():
            i, t = timed(lambda: a[0].replay(), steps)
            print('graph U1 alone (norms + clip + SGD): issue %.3f ms, %.3f ms per replay' % (i, t))
            if a[1] is not None:
                i, t = timed(lambda: a[1].replay(), steps)
                print('graph U2 alone (operand rebuild): issue %.3f ms, %.3f ms per replay' % (i, t))
        for k, a in g.V.items():
            i, t = timed(lambda: a.replay(), steps)
            print('graph V alone (VGG-16 forward): issue %.3f ms, %.3f ms per replay' % (i, t))
    tr.flush()
    torch.cuda.synchronize()


if __name__ == '__main__':
    main()
