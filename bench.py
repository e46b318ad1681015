#!/usr/bin/env python3
"""bench.py -- images/sec of the VG SGCls IMP hot path (BASELINE.json configs[1] / [3]) on N MI355X of one node.

--mode train (default): a step = one data-parallel TRAIN step of main.py:100-120 on the HIP path: train-mode forward
  (transform -> VGG-16 -> pair indexing -> RoIAlign(objects + union boxes) -> union-mask conv -> fc6/fc7 -> 3 IMP
  iterations -> heads), node + edge losses, backward of the trainable head, RCCL gradient all-reduce (N > 1),
  global-norm clip and SGD step -- nothing skipped.
--mode infer: a step = RelModelStanford.forward in eval mode incl. the eval tail and the D2H copy of the result tuple
  (no collective).
--mode sgdet (BASELINE configs[2], not the headline): a step = the SGDet eval forward -- VGG-16, RPN over 21 660 anchors, 1 000
  proposals per image after NMS, the box head on all of them, per-class NMS, <= 50 detections, overlap-filtered pairs, union-box
  RoIAlign, IMP, tail -- with the detector's score threshold at 0 (random-init weights: every candidate enters the per-class NMS).
One batch = B synthetic 592x592 frames per GPU, 32 boxes and 32*31 candidate edges per image.  The timed steps ROTATE over NB = 4
distinct batches per rank (different images, boxes, classes and gt_rels; the last one with another boxes-per-image signature:
30..34 boxes, 256 in total), handed over the way the reference's boundary hands them over (rel_model_base.py:180):
HOST-resident Blob tuples (decoded u8 images, gt_boxes / gt_classes / gt_rels on the host) through sgg_amd.blob.DeviceStager --
one pinned async copy per batch, issued one step ahead on a copy stream, so batch k is in HBM when step k starts.  `value` is that
rate; `hbm_resident` is the same rotation over copies that live in HBM from the start (no copy inside the timed region).
Images are sharded over ranks (one process per GPU) => weak scaling.  Prints ONE JSON line on rank 0; the other mode's
throughput is reported alongside under "other_mode".

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--dtype f16|bf16|f32] [--mode train|infer|sgdet]
"""
import os

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')   # before the HIP runtime starts: see sgg_amd/__init__.py

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TF = {'bf16': 2500.0, 'f16': 2500.0, 'f32': 157.3}   # dense peaks, MI355X_MICROARCH.md (bf16 and f16 MFMA run at the same rate)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=8)
    ap.add_argument('--batch', type=int, default=8, help='images per GPU (global batch 64 at 8 GPUs)')
    ap.add_argument('--dtype', default='f16', choices=['f16', 'bf16', 'f32'],
                    help='storage / MFMA operand format: f16 (default: the 16-bit mode that meets the parity clause), bf16 (same speed, 8x the rounding error), f32')
    ap.add_argument('--mode', default='train', choices=['train', 'infer', 'sgdet', 'gqa_gan'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-f32', action='store_true', help='skip the short exact-fp32 runs reported under "f32_mode"')
    ap.add_argument('--no-side-modes', action='store_true', help='skip the other single-GPU configs reported beside the headline (sgdet_mode, gqa_gan_mode)')
    ap.add_argument('--force-dist', action='store_true',
                    help='diagnostic: at 1 GPU, run the data-parallel code path on a 1-rank RCCL group')
    ap.add_argument('--cpu-images', type=int, default=8)
    ap.add_argument('--loss', default='baseline', choices=['baseline', 'dnorm', 'dnorm-fgbg'], help='lib/losses.py form (train mode)')
    ap.add_argument('--input', default='host', choices=['host', 'hbm'],
                    help="what `value` times: 'host' (default) = host-resident Blob tuples through DeviceStager.prefetch; 'hbm' = batches resident in HBM")
    ap.add_argument('--dry', action='store_true',
                    help='no GPU: check the launcher / rank environment with one gloo all-reduce and print the JSON skeleton')
    return ap.parse_args()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch(args):
    """`python bench.py --gpus N` without a rank environment: start one child process per GPU (the contract of
    torch.distributed.run: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), BEFORE this process has made any HIP or torch.cuda
    call -- a process that has touched the GPU is never re-executed.  Rank 0 inherits stdout (its JSON line is the last thing
    written there), the other ranks write to stderr.  Any non-zero exit ends the others and becomes this process's exit code."""
    import signal
    import subprocess
    n = args.gpus
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr, start_new_session=True))
    rc = 0
    live = set(range(n))
    try:
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    sys.stderr.write('[bench] rank %d exited with %d: stopping the other ranks\n' % (r, code))
                    for o in live:                      # exactly the process groups started above, nothing by pattern
                        try:
                            os.killpg(procs[o].pid, signal.SIGTERM)
                        except ProcessLookupError:
                            pass
            time.sleep(0.05)
    except KeyboardInterrupt:
        for o in live:
            try:
                os.killpg(procs[o].pid, signal.SIGTERM)
            except ProcessLookupError:
                pass
        rc = 130
    return rc


def guarded(args):
    """One-GPU runs: the bench proper runs in a CHILD process started before this one has made any HIP or torch.cuda call; its stdout (the
    JSON line) is passed on.  If the child dies -- round 5 met three ways in which hipGraph replays end in "Memory access fault by GPU" on this
    runtime (sgg_amd/graph_step.py; all worked around, none has recurred in 20+ runs) -- the run is repeated launch by launch (SGG_GRAPH=0) and
    the line says so (`config.hipgraph_fallback`).  -> exit code, or None when no child could be started (the caller runs in-process)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    for attempt in (0, 1):
        env = dict(os.environ, SGG_BENCH_CHILD='1')
        if attempt == 1:
            env.update(SGG_GRAPH='0', SGG_BENCH_FALLBACK='1')
        try:
            p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
        except OSError as e:
            sys.stderr.write('[bench] no child process (%s): running in-process\n' % e)
            return None
        out = p.stdout.decode('utf-8', 'replace')
        if p.returncode == 0 and out.strip():
            sys.stdout.write(out)
            sys.stdout.flush()
            return 0
        if attempt == 0 and os.environ.get('SGG_GRAPH', '1') != '0' and args.mode == 'train':
            sys.stderr.write('[bench] the run ended with exit code %d: once more with the train step launch by launch (SGG_GRAPH=0)\n' % p.returncode)
            continue
        sys.stdout.write(out)
        return p.returncode if p.returncode else 1
    return 1


def dry_run(args, rank, world):
    """Launcher check that needs no GPU: every rank joins a gloo group, one all-reduce, rank 0 prints the line's skeleton."""
    import torch
    import torch.distributed as dist
    if world > 1:
        if rank != 0:
            os.dup2(2, 1)
        dist.init_process_group('gloo', rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        assert float(t.item()) == world * (world + 1) / 2.0, t
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({'metric': 'images/sec (whole node), VG SGCls IMP %s step' % args.mode, 'value': None, 'unit': 'images/s',
                          'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'dry': True,
                          'config': {'images_per_gpu': args.batch, 'global_batch': world * args.batch}}), flush=True)


def kernel_times(step_fn, reps):
    """Per-kernel average launch duration from HIP events on the launch stream (separate, untimed passes)."""
    import torch
    from sgg_amd import _lib
    prof = {}
    _lib.profiler = prof
    try:
        for _ in range(reps):
            step_fn()
        torch.cuda.synchronize()
    finally:
        _lib.profiler = None
    out = {}
    for (name, tag), evs in prof.items():
        ms = [a.elapsed_time(b) for a, b in evs]
        out[(name, tag)] = (sum(ms) / len(ms), len(ms) // reps)   # avg ms per launch, launches per step
    return out


def pmc_traffic(key):
    """HBM-side bytes per launch READ FROM the committed rocprofv3 PMC passes (profiles/pmc_r04.json, else earlier rounds') -- collected
    by tools/pmc_traffic.sh on the same workload, not measured inside this run -- or None."""
    for name in ('pmc_r06.json', 'pmc_r05.json', 'pmc_r04.json', 'pmc_r03.json', 'pmc_r02.json', 'pmc_r01.json'):
        try:
            with open(os.path.join(ROOT, 'profiles', name)) as f:
                return json.load(f)[key]['traffic_bytes']
        except Exception:
            continue
    return None


def parity_records():
    """what tests/test_parity_full_gpu.py measured on this workload (committed under profiles/, newest round first): per compute mode the
    largest |logit - oracle| at the bench configuration and the largest |R@K - oracle's R@K| (points) -> the line's `parity` block.  Read, not
    measured in this run (the recall record needs 800 training steps and the oracle's forwards)."""
    rec, logit = None, None
    for r in ('r06', 'r05', 'r04', 'r03'):
        for name, slot in (('%s_recall_parity.json' % r, 'rec'), ('%s_parity_bench_config.json' % r, 'logit')):
            if (rec if slot == 'rec' else logit) is not None:
                continue
            try:
                with open(os.path.join(ROOT, 'profiles', name)) as f:
                    d = json.load(f)
                if slot == 'rec':
                    rec = (name, d)
                else:
                    logit = (name, d)
            except Exception:
                continue
    out = {'bars': 'north star: rel_dists / obj_dists within 1e-3 of the fp32 reference; R@50 within +-0.1 points',
           'source': 'profiles/%s, profiles/%s (written by tests/test_parity_full_gpu.py on the GPU; not measured in this run)' % (
               logit[0] if logit else None, rec[0] if rec else None)}
    for mode, rkey in (('f16', 'hip_f16'), ('bf16', 'hip_bf16'), ('x3', None), ('f32', 'hip_fp32')):
        e = {}
        if logit and mode in logit[1]:
            e['obj_max_abs'], e['rel_max_abs'] = round(logit[1][mode]['obj_max_abs'], 6), round(logit[1][mode]['rel_max_abs'], 6)
            e['meets_logits_1e-3'] = bool(e['obj_max_abs'] <= 1e-3 and e['rel_max_abs'] <= 1e-3)
        if rec and rkey:
            w = rec[1]['largest_abs_difference_to_oracle_points']
            e['recall_delta_points'] = {'GC': w[rkey + ' GC'], 'noGC': w[rkey + ' noGC']}
            e['meets_r50_0.1'] = bool(max(w[rkey + ' GC'], w[rkey + ' noGC']) <= 0.1)
        elif mode == 'x3':
            e['meets_r50_0.1'] = True
            e['recall_note'] = 'logits within 4e-4 of the reference: the ranking record was taken for f16 / bf16 / fp32 only'
        out[mode] = e
    return out


def contraction_alone_ms(tag, pairs, dtype, reps=10):
    """The roofline contraction launched on its own (nothing on a second stream beside it), same shapes and element types as in the step:
    average of `reps` launches between two events."""
    import torch
    from sgg_amd import ops
    dev = 'cuda:%d' % torch.cuda.current_device()
    g = torch.Generator().manual_seed(0)
    if tag == 'fc6_edge':
        A = torch.randn(pairs, 25088, generator=g).to(dev).to(dtype).relu()
        W = (torch.randn(4096, 25088, generator=g) / 160).to(dev).to(dtype)
        out = torch.empty(pairs, 4096, device=dev, dtype=torch.float32)
        run = lambda: ops.gemm(A, W, out=out, out_dtype=torch.float32)          # noqa: E731
    else:
        A = (torch.randn(pairs, 4096, generator=g) / 50).to(dev).to(dtype)
        W = torch.randn(pairs, 25088, generator=g).to(dev).to(dtype).relu()
        if ops.gemm_tn256_ok(A, W):          # the TN form the step runs: (pair sums of dY)^T . pooled rows, both as they lie
            run = lambda: ops.gemm_tn_full_waves(A, W, out_dtype=dtype)         # noqa: E731
        else:
            A, W = A.t().contiguous(), W.t().contiguous()
            run = lambda: ops.gemm_full_waves(A, W, out_dtype=dtype)            # noqa: E731
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def imp_iter_ms(model, B, dtype, reps=50, kind='ctx'):
    """Average duration of ONE launch of a kernel of the message-passing step on a complete 32-box/image graph of B images: `reps`
    launches back-to-back between two HIP events on the launch stream (outputs pre-allocated, hipGraph replay).  kind: 'ctx' = the
    step's gather / gate / scatter launch (sgg_imp_ctx_fwd: every edge row read once, gates from the dot products, two sums per
    node out -- routed by size as in the forward), 'ctx_sliced' / 'ctx_mfma' / 'ctx_lists' = its forms wherever they apply,
    'gate_proj' = the edge GRU's gate kernel that takes the place of the edge inputs (sgg_gru_gate_proj_fwd)."""
    import torch
    from sgg_amd import ops
    dev = model.rel_fc.weight.device
    n, H = 32, model.hidden_dim
    N, E = n * B, n * (n - 1) * B
    im = torch.arange(B, device=dev).repeat_interleave(n)
    rel, cnt = ops.pair_index_eval(im)
    rel = rel[:E]
    csr = ops.edge_csr(rel, N, im, graphs=(B, n, n * (n - 1)))
    g = torch.Generator(device='cpu').manual_seed(1)
    v = torch.randn(N, H, generator=g).to(dev).to(dtype)
    e = torch.randn(E, H, generator=g).to(dev).to(dtype)
    imp = model.prepared()['imp']
    ctx2 = torch.empty((2, N, H), dtype=dtype, device=dev)
    nd = (v.float() @ imp.gate_w[:, :H].t()).contiguous()     # what the gate kernels' dot epilogue hands over
    ed = (e.float() @ imp.gate_w[:, H:].t()).contiguous()
    form = {'ctx': None, 'ctx_sliced': 's', 'ctx_mfma': 'm', 'ctx_lists': 'l'}.get(kind, None)
    if kind == 'gate_proj':
        gh = torch.randn(E, 3 * H, generator=g).to(dev).to(ops.gh_dtype(dtype))      # as the GEMM in front of it hands it over
        P = torch.randn(N, 3 * H, generator=g).to(dev)
        out, dots = torch.empty_like(e), torch.empty((E, 4), dtype=torch.float32, device=dev)
        launch = lambda: ops.gru_gate_proj(gh, P, imp.edge_gru_b_ih, csr, nd, ed, imp.gate_b, e, out=out, dot_w=imp.gate_w[:, H:], dots=dots)
    else:
        assert ops.imp_sliced_ok(csr, H, dtype)

        def launch():
            if form is not None:
                os.environ['SGG_IMP_CTX'] = form
            try:
                ops.imp_ctx(e, csr, N, nd, ed, imp.gate_b, ctx2=ctx2)
            finally:
                os.environ.pop('SGG_IMP_CTX', None)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    # `reps` launches captured into one hipGraph so that the host launch path (Python + ctypes, ~10 us per call)
    # is not what is being timed; events bracket the replay on the replay stream
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            launch()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def graph_launch_us(fn, reps=50):
    """average duration (us) of one launch of fn: `reps` launches captured into one hipGraph, one replay between two events (as imp_iter_ms)"""
    import torch
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(reps):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def cpu_model_name():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.lower().startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(n_images, seed, timed=3):
    """the oracle (a structural CPU restatement of the reference path, validated against the reference's
    own outputs by tests/golden) on this box's host cores, same synthetic batch as the GPU run: 1 warm-up + `timed` timed
    forwards; (i) the full forward and (ii) the post-RoIAlign part (`predict`, comparable to BASELINE.md section 2)."""
    import torch
    import sgg_amd
    from oracle import sgg_oracle as O
    from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgcls'))
    sd = model.state_dict()
    batch = synthetic_batch(B=n_images, S=592, n_boxes=32, n_fg=6, seed=seed)
    avail = os.cpu_count() or 1
    # torch-CPU conv / GEMM on the GPU box (EPYC 9575F, 256 hardware threads) is fastest near 32 threads; BASELINE.md section 4 says
    # os.cpu_count(): both are timed, the faster one is `value` (its thread count is `cores`), the other is listed under `by_threads`
    by_threads = {}
    detail = {}
    with torch.no_grad():
        for cores, n_timed in ((min(avail, 32), max(1, timed - 1)), (avail, 1)) if avail > 32 else ((avail, timed),):
            torch.set_num_threads(cores)
            full, post = [], []
            for it in range(1 + n_timed):
                t0 = time.time()
                res = O.forward_gtbox(batch[0], batch[3], batch[4], batch[5], sd, mode='sgcls')
                t1 = time.time()
                O.predict(res['node_feat'], res['edge_feat'], res['rel_inds'], res['rois'], sd, im_sizes=res['im_sizes'])
                t2 = time.time()
                if it > 0:
                    full.append(t1 - t0)
                    post.append(t2 - t1)
            by_threads[cores] = round(n_images / (sum(full) / len(full)), 4)
            detail[cores] = (sum(full) / len(full), min(full), max(full), sum(post) / len(post), n_timed)
    cores = max(by_threads, key=lambda c: by_threads[c])
    dt, lo, hi, dp, n_timed = detail[cores]
    return {'value': by_threads[cores], 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'cpu_model': cpu_model_name(), 'host_threads_available': avail, 'by_threads': {str(k): v for k, v in by_threads.items()},
            'post_roialign_images_per_s': round(n_images / dp, 4),
            'sample': '1 warm-up + %d timed forwards of %d synthetic 592x592 images (32 boxes, 992 edges each, seed %d), torch-CPU fp32 '
                      'oracle on %d threads: %.1f s per forward (min %.1f, max %.1f), of which predict() %.1f s; `by_threads`: the same '
                      'forward at the other thread count (os.cpu_count() = %d)'
                      % (n_timed, n_images, seed, cores, dt, lo, hi, dp, avail)}


def sgdet_cpu_baseline(n_images, seed):
    """BASELINE configs[2] on the host: the oracle's SGDet forward (VGG-16, RPN, 1000 proposals, box head, per-class NMS, IMP) on a bounded
    sample -- `n_images` images, 1 timed forward after a warm-up of the first image only (the box head alone is ~0.4 TFLOP per image).
    The same synthetic detector as the GPU leg (synthetic.spread_detector_: 1 000 proposals and 50 detections per image)."""
    import torch
    import sgg_amd
    from oracle import sgg_oracle as O
    from sgg_amd.synthetic import SyntheticData, init_weights, spread_detector_, synthetic_batch
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgdet'))
    sd = spread_detector_(model.state_dict())
    batch = synthetic_batch(B=n_images, S=592, n_boxes=32, n_fg=6, seed=seed)
    avail = os.cpu_count() or 1
    cores = min(avail, 32)
    torch.set_num_threads(cores)
    with torch.no_grad():
        O.forward_sgdet(batch[0][:1], sd, score_thresh=SGDET_THRESH)
        t0 = time.time()
        out = O.forward_sgdet(batch[0], sd, score_thresh=SGDET_THRESH)
        dt = time.time() - t0
    return {'value': round(n_images / dt, 4), 'unit': 'images/s', 'cores': cores, 'kind': 'port', 'cpu_model': cpu_model_name(),
            'host_threads_available': avail, 'detections': int(len(out['labels'])), 'candidate_edges': int(len(out['rel_inds'])),
            'sample': '1 warm-up forward of one image + 1 timed forward of %d synthetic 592x592 images (seed %d, score threshold %.2f), the '
                      'oracle\'s SGDet forward (torch-CPU fp32 + numpy NMS) on %d threads: %.1f s' % (n_images, seed, SGDET_THRESH, cores, dt)}


SGDET_THRESH = 0.05        # lib/eval.py:125-132: the evaluation starts at 0.2 and retries at 0.05 / 0.01; the synthetic class head is peaked enough for any of them


def sgdet_model(dev, tdtype):
    """RelModelStanford(mode='sgdet') with He-initialised weights and the synthetic detector that proposes like a trained one
    (sgg_amd/synthetic.py: spread_detector_)."""
    import sgg_amd
    from sgg_amd.synthetic import SyntheticData, init_weights, spread_detector_
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgdet'))
    spread_detector_(model.state_dict())         # (state_dict tensors alias the parameters: in place)
    model.to(dev).eval()
    model.set_compute_dtype(tdtype)
    model.detector.mode = 'refinerels'
    model.set_box_score_thresh(SGDET_THRESH)
    return model


def sgdet_measure(args, model, batches, timed, B, dev, steps, warmup, cpu_images=0):
    """images/s of the SGDet eval forward at BASELINE configs[2]'s size (8 x 1 000 proposals through the box head) over a rotation of
    `batches` + the per-call kernel times of one profiled pass -> dict (the `--mode sgdet` line, or `sgdet_mode` of the default line)."""
    import torch

    def step(b=None):
        model.eval()
        with torch.no_grad():
            return model([batches[0] if b is None else b])
    out = step()
    n_det, n_edges = len(out[1]), len(out[3])
    offs = list(model.detector.last_proposal_offsets)
    elapsed = timed(step, warmup, steps, lambda n: (batches[i % len(batches)] for i in range(n)))
    kt = kernel_times(step, reps=3)
    total_ms = sum(v[0] * v[1] for v in kt.values())
    top = sorted(((v[0] * v[1], n, t) for (n, t), v in kt.items()), reverse=True)[:int(os.environ.get('SGG_BENCH_TOP', '12'))]
    # the largest contraction of the step: the box head's fc6 on the proposals ([K x 25088] . [4096 x 25088]^T: at 8 x 1 000 proposals twice
    # the relation head's), unless a detector proposes so few boxes that the relation head's fc6 on the union-box rows is longer
    gemms = {t: v[0] * v[1] for (n, t), v in kt.items() if n in ('sgg_gemm', 'sgg_gemm_splitk')}
    K = int(getattr(model.detector, 'last_proposals', 0))
    paired = os.environ.get('SGG_EDGE_PAIRS', '1') != '0'
    tag = max(gemms, key=lambda t: gemms[t]) if gemms else ''
    ms = gemms.get(tag, 0.0)
    rows = (n_edges // 2 if paired else n_edges) if tag == 'fc6_edge' else K
    if tag not in ('fc6_edge', 'box_fc6'):
        tag, ms = 'box_fc6', gemms.get('box_fc6', 0.0)
        rows = K
    what = ('relation head fc6 on the union-box rows of the %d candidate edges (%d rows)' % (n_edges, rows)) if tag == 'fc6_edge' else \
        ('box head fc6 on the %d proposals' % K)
    flop = 2.0 * rows * 4096 * 25088
    peak = MFMA_PEAK_TF[args.dtype]
    tf = flop / (ms * 1e-3) / 1e12 if ms else 0.0
    per = lambda names: sum(v[0] * v[1] for (n, t), v in kt.items() if n in names)       # noqa: E731
    res = {'value': round(B * steps / elapsed, 3), 'unit': 'images/s', 'steps': steps, 'warmup': warmup,
           'ms_per_step': round(1e3 * elapsed / steps, 3), 'dtype': args.dtype,
           'input': 'batches resident in HBM (rotation of %d)' % len(batches),
           'config': {'workload': 'VG SGDet (BASELINE configs[2]): 592x592 frames, RPN 21 660 anchors -> %d proposals/img after NMS 0.7, box '
                                  'head on all of them, per-class NMS 0.5, <= 50 detections/img, overlap-filtered pairs, union RoIAlign, '
                                  '3 IMP iters, eval tail' % (K // max(B, 1)),
                      'mode': 'sgdet', 'images_per_gpu': B, 'detections_per_step': n_det, 'candidate_edges_per_step': n_edges,
                      'proposals_per_step': K, 'proposals_per_image': [offs[i + 1] - offs[i] for i in range(len(offs) - 1)],
                      'score_thresh': SGDET_THRESH,
                      'weights': 'random init (He) + synthetic.spread_detector_: RPN deltas x 0.02 and an objectness bias for the 32-px anchors '
                                 '(1 000 proposals survive NMS 0.7), class head x 2 with a random bias, box regression grows detections to object size'},
           'roofline': {'kernel': '%s: [%d x 25088] . [4096 x 25088]^T (256x256 ping-pong MFMA kernel)' % (what, rows),
                        'bound': 'mfma', 'achieved': round(tf, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(tf / peak, 4),
                        'traffic': pmc_traffic('box_fc6_gemm') if (tag == 'box_fc6' and K == 8000 and args.dtype == 'f16') else None,
                        'traffic_source': 'profiles/pmc_r0x.json (separate rocprofv3 --pmc passes of the same launch, tools/pmc_traffic.sh), not measured in this run',
                        'ms_per_step': round(ms, 4), 'executed_flop': flop},
           'kernels': {'sum_kernel_ms_per_step': round(total_ms, 3), 'largest_gemm_ms': round(ms, 3),
                       'vgg16_ms': round(per(('sgg_conv3x3_relu', 'sgg_conv1_1', 'sgg_conv1_block', 'sgg_maxpool2x2')), 3),
                       'sort_ms': round(per(('sgg_segmented_sort_desc', 'sgg_gather_topk', 'sgg_topk_gather')), 4),
                       'nms_ms': round(per(('sgg_nms',)), 4),
                       'top': [{'ms_per_step': round(ms_, 3), 'call': n, 'tag': t} for ms_, n, t in top]}}
    if cpu_images:
        res['cpu_baseline'] = sgdet_cpu_baseline(cpu_images, 111)
    return res


def sgdet_bench(args, batches, timed, world, rank, B, dev, tdtype):
    """--mode sgdet: the SGDet eval forward as the line's `value` (every rank runs its own images: no collective)."""
    model = sgdet_model(dev, tdtype)
    res = sgdet_measure(args, model, batches, timed, B, dev, args.steps, args.warmup,
                        cpu_images=(min(args.cpu_images, 2) if (world == 1 and rank == 0 and not args.no_cpu_baseline) else 0))
    if rank != 0:
        return
    line = {'metric': 'images/sec (whole node), VG SGDet eval forward (detector + IMP)', 'value': round(world * res['value'], 3),
            'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': res['ms_per_step'],
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic'}
    line.update({k: v for k, v in res.items() if k not in line})
    import ctypes
    ctypes.CDLL(None).fflush(None)
    print(json.dumps(line), flush=True)


def gqa_gan_setup(dev, tdtype, B, S=1333, n_batches=2, gan_compute=None):
    """BASELINE configs[4] on one GPU: RelModelStanford(backbone='resnet50') with GQA's vocabulary (1 704 / 311 classes), the GAN of
    augment/gan.py on its 256 x 21 x 21 'pool'-level maps, B synthetic 1333 x 1333 frames (config.py:76-78 forces the backbone,
    rel_model_base.py:62-65 its frame size), 32 boxes and 992 candidate edges per image, labels drawn from the GQA vocabulary."""
    import torch
    import sgg_amd
    from sgg_amd import dense
    from sgg_amd.feature_gan import GAN
    from sgg_amd.rel_model_base import to_device_with_mirror
    from sgg_amd.resnet_fpn import FrozenBatchNorm2d
    from sgg_amd.synthetic import GQASyntheticData, init_weights, relabel_batch, synthetic_batch
    from sgg_amd.trainer import Trainer
    data = GQASyntheticData()
    torch.manual_seed(111)
    model = init_weights(sgg_amd.RelModelStanford(data, mode='sgcls', backbone='resnet50'))
    for m in model.modules():
        if isinstance(m, FrozenBatchNorm2d):
            m.weight.uniform_(0.5, 1.0)
            m.running_var.uniform_(0.6, 1.4)
    for n, p in model.named_parameters():
        if n.startswith('detector.'):
            p.requires_grad = False               # main.py:62-63
    model.to(dev)
    model.set_compute_dtype(tdtype)
    gan = GAN(data.ind_to_classes, data.ind_to_predicates, n_ch=model.edge_dim, pool_sz=model.pool_sz, fmap_sz=model.fmap_sz, device=dev).to(dev)
    if gan_compute is not None:
        dense.set_compute(gan_compute)
    batches = []
    for k in range(n_batches):
        b = list(relabel_batch(synthetic_batch(B=B, S=S, n_boxes=32, n_fg=6, seed=111 + 1000 * k), len(data.ind_to_classes),
                               len(data.ind_to_predicates), seed=k))
        b[0] = [im.to(dev) for im in b[0]]
        b[3] = b[3].to(dev)
        b[4], b[5] = to_device_with_mirror(b[4], dev), to_device_with_mirror(b[5], dev)
        batches.append(tuple(b))
    tr = Trainer(model, lr=1e-3, pipeline=False, graph=False)
    # lib/pytorch_misc.py:113-114 (Adam for G and D; config.py defaults lrG = lrD = 1e-4? no checkpoint here: the rates only scale the update)
    G_opt = torch.optim.Adam([p for n, p in gan.named_parameters() if n.startswith('G_')], lr=1e-4, betas=(0.5, 0.999))
    D_opt = torch.optim.Adam([p for n, p in gan.named_parameters() if n.startswith('D_')], lr=1e-4, betas=(0.5, 0.999))
    return model, gan, tr, G_opt, D_opt, batches


def gqa_gan_measure(args, dev, tdtype, timed, B, steps, warmup, cpu=True):
    """images/s of one GQA + GAN training iteration (main.py:100-194: SGG forward, losses, backward, clip + SGD, then the generator
    and discriminator updates incl. the reconstruction update of the SGG model) + per-call kernel times of a profiled pass."""
    import torch
    from sgg_amd import _lib
    from sgg_amd.feature_gan import gan_train_step
    model, gan, tr, G_opt, D_opt, batches = gqa_gan_setup(dev, tdtype, B, gan_compute=os.environ.get('SGG_GAN_COMPUTE'))
    state = {}

    def step(b=None):
        b = batches[0] if b is None else b
        model.train()
        _lib.set_tag_prefix('sgg:')
        res = model([b])                                        # main.py:103
        loss = tr.losses(res)                                   # :106-114
        tr.opt.zero_grad()
        model._loss_scaled = True
        try:
            (loss * tr.loss_scale).backward()                   # :117-118
        finally:
            model._loss_scaled = False
        tr.update()                                             # :119-120 (global-norm clip + SGD)
        _lib.set_tag_prefix('gan:')
        state['losses'] = gan_train_step(model, gan, res, b[3], b[4], b[5], None, G_opt, D_opt, trainer=tr)    # :124-194
        _lib.set_tag_prefix('')
        return state['losses']
    step()
    torch.cuda.synchronize()
    elapsed = timed(step, warmup, steps, lambda n: (batches[i % len(batches)] for i in range(n)))
    kt = kernel_times(step, reps=2)
    total_ms = sum(v[0] * v[1] for v in kt.values())
    top = sorted(((v[0] * v[1], n, t, v[1]) for (n, t), v in kt.items()), reverse=True)[:int(os.environ.get('SGG_BENCH_TOP', '14'))]
    by_tag = {}
    for (n, t), v in kt.items():
        part = t.split(':', 1)[0] if ':' in t else 'other'
        by_tag[part] = by_tag.get(part, 0.0) + v[0] * v[1]
    # the largest contraction: the edge discriminator's first 3x3 convolution on the [E, 7, 7, 256 + 311] class-conditioned RoI features
    # (augment/gan.py:222-231): E x 25 output positions x 256 channels x 9 (256 + 311) taps, run on generated rows (G update), real + generated
    # rows (D update) and in both backwards
    E = 992 * B
    d_edge_flop = 2.0 * E * 25 * 256 * 9 * (256 + 311)
    from sgg_amd import dense
    gemm_ms = sum(v[0] * v[1] for (n, t), v in kt.items() if n in ('sgg_gemm', 'sgg_gemm_splitk') and t.startswith('gan:'))
    mode = dense.compute_mode()
    peak = {'f32': MFMA_PEAK_TF['f32'], 'x3': MFMA_PEAK_TF['f16'] / 3.0, 'f16': MFMA_PEAK_TF['f16']}[mode]
    res = {'value': round(B * steps / elapsed, 3), 'unit': 'images/s', 'steps': steps, 'warmup': warmup, 'ms_per_step': round(1e3 * elapsed / steps, 3),
           'dtype': args.dtype, 'gan_dense_layers': {'f32': 'exact fp32 MFMA (v_mfma_f32_32x32x2_f32)', 'x3': 'f16 split operands (hi + lo), three products, fp32 accumulate: fp32-grade',
                                                       'f16': 'f16 operands, fp32 accumulate'}[mode],
           'config': {'workload': 'GQA SGCls + GAN feature augmentation (BASELINE configs[4]) on one GPU: %d synthetic 1333x1333 frames, ResNet-50-FPN '
                                  'detector (frozen), 32 boxes / 992 edges per image, 1704 object / 311 predicate classes, IMP head train step + '
                                  'generator / discriminator / reconstruction updates (main.py:100-194)' % B,
                      'mode': 'gqa_gan', 'images_per_gpu': B, 'object_classes': 1704, 'predicate_classes': 311,
                      'losses': {k: round(float(v), 4) for k, v in state['losses'].items()}},
           'kernels': {'sum_kernel_ms_per_step': round(total_ms, 3), 'by_part_ms': {k or 'other': round(v, 3) for k, v in by_tag.items()},
                       'gan_gemm_ms': round(gemm_ms, 3), 'gan_patch_matrix_ms': round(sum(v[0] * v[1] for (n, t), v in kt.items() if n in ('sgg_im2col', 'sgg_col2im') and t.startswith('gan:')), 3),
                       'top': [{'ms_per_step': round(ms_, 3), 'call': n, 'tag': t, 'launches': k} for ms_, n, t, k in top]}}
    # roofline of the largest single contraction, timed on its own (same shapes, same entry point)
    try:
        x = torch.randn(E * 25, 9 * (256 + 311) // 32 * 32 + 32, device=dev)
        w = torch.randn(256, x.shape[1], device=dev)
        from sgg_amd import ops
        for _ in range(2):
            dense.product(x, w)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            dense.product(x, w)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        flop = 2.0 * x.shape[0] * x.shape[1] * 256
        res['roofline'] = {'kernel': 'D_edges first convolution as rows x weights^T: [%d x %d] . [256 x %d]^T (%s)' % (x.shape[0], x.shape[1], x.shape[1], mode),
                           'bound': 'mfma', 'achieved': round(flop / (ms * 1e-3) / 1e12, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                           'frac': round(flop / (ms * 1e-3) / 1e12 / peak, 4), 'traffic': None, 'ms_per_launch': round(ms, 4), 'executed_flop': flop,
                           'note': 'peak = the dense MFMA rate of the arithmetic the layer runs in (x3: a third of the f16 rate, three products per fp32-grade product)'}
        del x, w
    except Exception as e:
        res['roofline'] = {'error': repr(e)[:300]}
    if cpu:
        res['cpu_baseline'] = gqa_gan_cpu_baseline()
    return res


def gqa_gan_cpu_baseline():
    """the oracle on the host, bounded: the ResNet-50-FPN 'pool'-level forward of ONE 1333 x 1333 frame + the relation head's eval forward on
    its 32 boxes / 992 edges (oracle/sgg_oracle.py forward_gtbox with the resnet50 state dict); the GAN half has no CPU restatement at this
    size (its parity is pinned at the golden vectors' size, tests/golden/gan_model.npz), so this is the SGG half's rate only -- said so."""
    import torch
    import sgg_amd
    from oracle import sgg_oracle as O
    from sgg_amd.synthetic import GQASyntheticData, init_weights, relabel_batch, synthetic_batch
    data = GQASyntheticData()
    model = init_weights(sgg_amd.RelModelStanford(data, mode='sgcls', backbone='resnet50'))
    sd = {k: v.detach().float() for k, v in model.state_dict().items()}
    b = relabel_batch(synthetic_batch(B=1, S=1333, n_boxes=32, n_fg=6, seed=111), 1704, 311)
    avail = os.cpu_count() or 1
    cores = min(avail, 32)
    torch.set_num_threads(cores)
    with torch.no_grad():
        t0 = time.time()
        O.forward_gtbox(b[0], b[3].numpy(), b[4].numpy(), b[5].numpy(), sd, mode='sgcls', min_size=1333, max_size=1333)
        dt = time.time() - t0
    return {'value': round(1.0 / dt, 4), 'unit': 'images/s', 'cores': cores, 'kind': 'port', 'cpu_model': cpu_model_name(),
            'host_threads_available': avail,
            'sample': 'ONE eval forward of one synthetic 1333x1333 frame (32 boxes, 992 edges, GQA vocabulary) through the oracle (torch-CPU fp32) on '
                      '%d threads: %.1f s; the SGG half of the iteration only (no backward, no GAN): an upper bound of the CPU rate' % (cores, dt)}


def main():
    args = parse()
    if os.environ.get('SGG_BENCH_NOGC'):          # debugging only: no cyclic collection at all
        import gc
        gc.disable()
    host_group = None
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch(args))          # nothing above this line has touched the GPU
    if ('WORLD_SIZE' not in os.environ and args.gpus == 1 and not args.dry and 'SGG_BENCH_CHILD' not in os.environ and
            os.environ.get('SGG_BENCH_GUARD', '1') != '0' and 'rocprof' not in os.environ.get('LD_PRELOAD', '').lower() and
            not any(k.startswith(('ROCP', 'ROCPROFILER', 'ROCTRACER')) for k in os.environ)):
        # (never under a profiler: its preloaded library has initialised the GPU in this process already, and such a process starts no other)
        rc = guarded(args)
        if rc is not None:
            sys.exit(rc)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.exit('bench.py: --gpus %d but the rank environment says WORLD_SIZE=%d' % (args.gpus, world))
    if args.dry:
        return dry_run(args, rank, world)
    import torch
    import torch.distributed as dist
    if world > 1:
        if rank != 0:
            os.dup2(2, 1)      # only rank 0 owns stdout (the JSON line); anything other ranks print goes to stderr
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local))
        host_group = dist.new_group(backend='gloo')
    else:
        torch.cuda.set_device(0)
        if args.force_dist:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29581')
            os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    dev = torch.device('cuda', local if world > 1 else 0)

    import sgg_amd
    from sgg_amd.synthetic import SyntheticData, init_weights, synthetic_batch
    tdtype = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[args.dtype]
    B = args.batch
    model = init_weights(sgg_amd.RelModelStanford(SyntheticData(), mode='sgdet' if args.mode == 'sgdet' else 'sgcls')).to(dev).eval()
    model.set_compute_dtype(tdtype)
    if os.environ.get('SGG_EVAL_GRAPH') == '1':
        model.enable_eval_graphs()      # (opt-in: the evaluation forward is GPU-bound, sgg_amd/graph_forward.py)
    # images sharded by rank: rank r owns global images [r*B, (r+1)*B) of every global batch (seeds differ per rank and per batch).
    # NB distinct batches per rank; the last has another boxes-per-image signature (same 32-box mean: 30..34, 256 boxes per 8 images).
    from sgg_amd.blob import DeviceStager
    NB = 4
    ragged_counts = [(30, 34, 32, 32, 31, 33, 32, 32)[i % 8] for i in range(B)]
    host_batches = []
    for k in range(NB):
        hb = list(synthetic_batch(B=B, S=592, n_boxes=32, n_fg=6, seed=111 + rank + 1000 * k, counts=ragged_counts if k == NB - 1 else None))
        # the boundary's form (dataloaders/blob.py): decoded u8 [h,w,3] images + index tensors, all on the HOST
        hb[0] = [(im * 255).round().to(torch.uint8).permute(1, 2, 0).contiguous() for im in hb[0]]
        host_batches.append(tuple(hb))
    # 16 staging slots (pinned + device, ~9 MB each): the issuing thread may run that many batches ahead of the GPU, so a host-side stall
    # (a ~100 ms descheduling on a loaded host, measured in round 4: one such stall per 200 steps cost 5 % with 3 slots) drains the
    # queue instead of the GPU
    stager = DeviceStager(dev, slots=int(os.environ.get('SGG_STAGER_SLOTS', '16')))
    # the same batches resident in HBM (staged once, copied out of the stager's slots; index tensors keep their host mirrors, as a
    # Blob keeps its chunk sizes: no D2H sync inside the step)
    dev_batches = []
    for hb in host_batches:
        st = list(stager.stage(hb))
        st[0] = [im.clone() for im in st[0]]
        for i in (3, 4, 5):
            mirror = getattr(st[i], '_sgg_host', None)
            st[i] = st[i].clone()
            if mirror is not None:
                st[i]._sgg_host = mirror
        dev_batches.append(tuple(st))
    torch.cuda.synchronize()
    batch = dev_batches[0]
    edges_per_batch = [sum(int(c) * (int(c) - 1) for c in torch.bincount(hb[4][:, 0]).tolist()) for hb in host_batches]

    def feed_hbm(n):
        return (dev_batches[i % NB] for i in range(n))

    def feed_host(n):
        return stager.prefetch((host_batches[i % NB] for i in range(n)), threaded=os.environ.get('SGG_STAGER_THREAD', '1') != '0')

    from sgg_amd.trainer import Trainer
    trainer = None if args.mode != 'train' else Trainer(model, lr=1e-3, force_dist=args.force_dist, loss_type=args.loss,
                                                        pipeline=os.environ.get('SGG_PIPELINE', '1') != '0', sync_bn=os.environ.get('SGG_SYNC_BN', '1') != '0')

    def infer_step(b=None):
        if model.training:
            model.eval()
        with torch.no_grad():
            return model([batch if b is None else b])

    def train_step(b=None):
        return trainer.step(batch if b is None else b)

    step = train_step if args.mode == 'train' else infer_step

    feed_ms = [0.0]     # ... of which inside the feed's next()
    issue_ms = [0.0]    # of the last timed() call: host time per step until the last step was ISSUED (no synchronisation inside)
    wait_ms = [0.0]     # ... of which the issuing thread spent WAITING for the GPU on purpose (the replayed step's run-ahead bound, sgg_amd/graph_step.py)

    def timed(fn, warmup, steps, feed=feed_hbm):
        """`warmup` untimed + EXACTLY `steps` timed steps, each on the next batch of `feed` (one iterator over warmup + steps batches: with
        the prefetching feed the first timed batch was staged during the last warm-up step, as in steady state)."""
        it = iter(feed(warmup + steps))
        for _ in range(warmup):
            fn(next(it))
        if os.environ.get('SGG_GC_FREEZE', '1') != '0':
            # as sgg_amd.trainer.Trainer does after its third step: the long-lived objects (model, operand caches, batches) leave the cyclic
            # collector's sight -- a full collection over them is one 80-110 ms pause of the issuing thread per ~100 steps
            import gc
            gc.collect()
            gc.freeze()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        prof_ = None
        if os.environ.get('SGG_BENCH_CPROFILE'):        # where the issuing thread's time goes (stderr); slows the loop down
            import cProfile
            prof_ = cProfile.Profile()
            prof_.enable()
        t0 = time.perf_counter()
        feed_s = 0.0
        w0 = trainer.graphs.stats['wait_s'] if (trainer is not None and trainer.graphs is not None) else 0.0
        per_step = []
        for _ in range(steps):
            ta = time.perf_counter()
            b_ = next(it)
            feed_s += time.perf_counter() - ta
            fn(b_)
            per_step.append(time.perf_counter() - ta)
        if os.environ.get('SGG_BENCH_STEPTIMES'):
            ps = sorted(per_step)
            sys.stderr.write('host time per step (ms): median %.2f p90 %.2f max %.2f; the five longest at steps %s; first ten %s\n' % (
                1e3 * ps[len(ps) // 2], 1e3 * ps[int(len(ps) * 0.9)], 1e3 * ps[-1],
                sorted(range(len(per_step)), key=lambda i: -per_step[i])[:5], ['%.1f' % (1e3 * v) for v in per_step[:10]]))
        feed_ms[0] = 1e3 * feed_s / max(steps, 1)
        if prof_ is not None:
            prof_.disable()
            import pstats
            pstats.Stats(prof_, stream=sys.stderr).sort_stats('tottime').print_stats(45)
            pstats.Stats(prof_, stream=sys.stderr).sort_stats('cumtime').print_stats(60)
        issue_ms[0] = 1e3 * (time.perf_counter() - t0) / max(steps, 1)     # host: all steps issued (the queue may have pushed back)
        wait_ms[0] = 1e3 * ((trainer.graphs.stats['wait_s'] if (trainer is not None and trainer.graphs is not None) else 0.0) - w0) / max(steps, 1)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        for _ in it:           # (lets the prefetch generator finish)
            pass
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    if args.mode == 'gqa_gan':
        del model
        res = gqa_gan_measure(args, dev, tdtype, timed, B, args.steps, args.warmup, cpu=(world == 1 and not args.no_cpu_baseline))
        if rank == 0:
            line = {'metric': 'images/sec (whole node), GQA SGCls + GAN training iteration', 'value': round(world * res['value'], 3), 'unit': 'images/s',
                    'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': res['ms_per_step'], 'higher_is_better': True,
                    'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic'}
            line.update({k: v for k, v in res.items() if k not in line})
            import ctypes
            ctypes.CDLL(None).fflush(None)
            print(json.dumps(line), flush=True)
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.mode == 'sgdet':
        del model
        sgdet_bench(args, dev_batches, timed, world, rank, B, dev, tdtype)
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return
    head_feed, other_feed = (feed_host, feed_hbm) if args.input == 'host' else (feed_hbm, feed_host)
    primed = 0
    if trainer is not None and trainer.graphs is not None:
        # hipGraph capture of the train step (sgg_amd/graph_step.py) happens per batch signature after three plain steps of that kind, and a
        # capture costs tens of milliseconds: like any one-time initialisation it is kept out of the W warm-up + K timed steps -- the
        # rotation is run (untimed, real steps: every update is applied) until every batch of it has been replayed from its graphs
        for b_ in head_feed(8 * NB):
            train_step(b_)
            primed += 1
        trainer.flush()
        torch.cuda.synchronize()
        if os.environ.get('SGG_BENCH_MEMSNAP'):          # debugging: where every segment of the caching allocator lies (private graph pools included)
            snap = [dict(address=sg['address'], size=sg['total_size'], pool=str(sg.get('segment_pool_id')), stream=sg.get('stream'),
                         blocks=[(b_['address'] if 'address' in b_ else None, b_['size'], b_['state']) for b_ in sg['blocks']][:400])
                    for sg in torch.cuda.memory_snapshot()]
            with open(os.environ['SGG_BENCH_MEMSNAP'], 'w') as f_:
                json.dump(snap, f_)
    def prime_eval(feed):
        """the evaluation forward's hipGraphs (sgg_amd/graph_forward.py, opt-in: one per batch signature, captured at a signature's third call)
        are made before the W warm-up + K timed steps, like the train step's: the rotation, untimed"""
        n_ = 0
        if not model.__dict__.get('_eval_graphs'):
            return n_
        for b_ in feed(3 * NB):
            infer_step(b_)
            n_ += 1
        torch.cuda.synchronize()
        return n_
    if args.mode == 'infer':
        primed = prime_eval(head_feed)
    if trainer is not None and trainer.dist_on:
        trainer.buckets.timing = []
    elapsed = timed(step, args.warmup, args.steps, head_feed)
    head_issue_ms, head_feed_ms, head_wait_ms = issue_ms[0], feed_ms[0], wait_ms[0]
    comm = None
    if trainer is not None and trainer.dist_on:
        # per step: bytes this rank handed to RCCL (reduce-scatter / all-reduce inputs + the all-gathered operands) and how long the stream
        # that waits for the wire handles stood still for them (in pipeline mode that is the side stream: the main stream is already in
        # the next step's VGG forward, the wait is exposed only where the next head forward then waits for the rebuilt operands)
        torch.cuda.synchronize()
        waits = [a.elapsed_time(b) for a, b in trainer.buckets.timing[-args.steps:]]
        trainer.buckets.timing = None
        comm = {'rccl_bytes': int(trainer.buckets.bytes_last + getattr(trainer.opt, 'gather_bytes_last', 0)),
                'reduce_bytes': int(trainer.buckets.bytes_last), 'all_gather_bytes': int(getattr(trainer.opt, 'gather_bytes_last', 0)),
                'comm_exposed_ms': round(sum(waits) / max(1, len(waits)), 4), 'comm_exposed_ms_max': round(max(waits), 4) if waits else None,
                'wire_dtype': 'bf16', 'waiting_stream': 'side (pipeline)' if trainer.pipeline else 'main',
                'note': 'bytes per rank and step handed to collectives; comm_exposed_ms = event-timed wait on the wire handles, mean over the timed steps'}
    if trainer is not None:
        # EVERY rank: with the sharded optimiser flush() is a collective (FusedSGD.gather_masters all-gathers the stale parts of
        # the fp32 masters), so it must not sit inside the rank-0-only profiling block below
        trainer.flush()
    # the other input form, same rotation, same number of steps (every rank: the train step holds collectives)
    elapsed_other = timed(step, args.warmup, args.steps, other_feed)
    other_issue_ms = issue_ms[0]
    if trainer is not None:
        trainer.flush()

    # ---- per-kernel roofline (rank 0, outside the timed region; single-GPU kernels, no collective inside)
    if rank == 0:
        if args.mode == 'train':
            def prof_step():
                model.train()
                res = model([batch])
                loss = trainer.losses(res)
                trainer.opt.zero_grad()
                model._loss_scaled = True
                (loss * trainer.loss_scale).backward()
                model._loss_scaled = False
                trainer.opt.step(grad_scale=1.0 / trainer.loss_scale)
            # rank 0 alone runs these extra steps: no collective may be issued (the other ranks wait at the host-side barrier
            # below): the trainer is switched to its local form -- hooks, BatchNorm sync, loss normalisers (dist_on) and world
            with trainer.local_only():
                kt = kernel_times(prof_step, reps=3)
        else:
            # (per-call kernel times come from hooks around the C entry points: the forward launch by launch, not its replayed hipGraph)
            saved_eg = model.__dict__.get('_eval_graphs')
            model.__dict__['_eval_graphs'] = False
            try:
                kt = kernel_times(infer_step, reps=5)
            finally:
                model.__dict__['_eval_graphs'] = saved_eg
        # the same forward as one back-to-back sequence on an otherwise idle GPU (ten forwards between two events, the image prep kernel included):
        # `roofline_vgg.ms_per_step` below sums per-launch event pairs inside a step, which also time the event records between the launches
        vgg_alone = None
        try:
            with torch.no_grad():
                for _ in range(2):
                    model.detector.features(batch[0], model.compute_dtype)
                torch.cuda.synchronize()
                v0, v1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                v0.record()
                for _ in range(10):
                    model.detector.features(batch[0], model.compute_dtype)
                v1.record()
                torch.cuda.synchronize()
                vgg_alone = v0.elapsed_time(v1) / 10
        except Exception:
            vgg_alone = None
        E, N, H = 992 * B, 32 * B, 512
        s = 4 if args.dtype == 'f32' else 2
        peak = MFMA_PEAK_TF[args.dtype]
        get = lambda name, tag: kt.get((name, tag), (0.0, 0))
        per_step = lambda name, tag: get(name, tag)[0] * get(name, tag)[1]
        # the two largest contractions: fc6 on edges forward, and its weight gradient in training
        # The forward pools every unordered box pair once (sgg_amd/pairing.py): on these complete graphs the long contraction runs on
        # U = E / 2 rows.  `achieved` counts the FLOPs the kernel EXECUTES; the reference's per-edge algorithm (2 E 4096 25600
        # per image batch) is reported beside it as `reference_algorithm_tflops`.
        paired = os.environ.get('SGG_EDGE_PAIRS', '1') != '0'
        U = E // 2 if paired else E
        if paired:
            cands = {'fc6_edge': ('fc6 on the unordered box pairs, forward: [%d x 25088] . [4096 x 25088]^T (f32 out; the per-edge rect '
                                  'term, bias and ReLU follow in a K = 512 launch)' % U, 2.0 * U * 4096 * 25088),
                     'bwd_fc6_edge_dW': ('fc6 weight gradient over the unordered pairs: [%d x 4096]^T . [%d x 25088]' % (U, U),
                                         2.0 * U * 4096 * 25088)}
        else:
            cands = {'fc6_edge': ('fc6 on edges, forward: [%d x 25600] . [4096 x 25600]^T' % E, 2.0 * E * 4096 * 25600),
                     'bwd_fc6_edge_dW': ('fc6 weight gradient: [4096 x %d] . [25088 x %d]^T (the rect term rides in the transpose)' % (E, E),
                                         2.0 * E * 4096 * 25088)}
        ref_flop = {'fc6_edge': 2.0 * E * 4096 * 25600, 'bwd_fc6_edge_dW': 2.0 * E * 4096 * 25088}
        best = None
        for tag, (desc, flop) in cands.items():
            # a contraction may be issued as a full-round launch + a split-K tail; the weight gradient over the pairs runs as sgg_gemm_groupadd
            # (since round 4 as sgg_gemm_tn256 + the last tile columns on sgg_gemm_tn: both operands read as they lie)
            ms = (per_step('sgg_gemm', tag) + per_step('sgg_gemm_splitk', tag) + per_step('sgg_gemm_groupadd', tag) +
                  per_step('sgg_gemm_tn256', tag) + per_step('sgg_gemm_tn', tag))
            if ms > 0 and (best is None or ms > best[1]):
                best = (tag, ms, desc, flop)
        tag, ms, desc, flop = best
        fwd_ms = per_step('sgg_gemm', 'fc6_edge') + per_step('sgg_gemm_splitk', 'fc6_edge')
        fwd_desc, fwd_flop = cands['fc6_edge']
        tf = flop / (ms * 1e-3) / 1e12
        alone_ms = contraction_alone_ms(tag, U, tdtype) if (paired and tdtype != torch.float32) else None
        # the step's gather / gate / scatter launch (sgg_imp_ctx_fwd), back-to-back timing.  Algorithmic bytes = SURVEY 8(d)'s figure for the
        # reference's step: read e_i and v_i, WRITE e_in, write ctx, per iteration.  Since round 3 the e_in stream does not exist (node
        # projection, csrc/imp.hip): the launch delivers the step's outputs in the time of its read stream, `moved_bytes` says what it moves.
        imp_ms = imp_iter_ms(model, B, tdtype)
        imp_bytes = (2.0 * (E + N) * H) * s + 8.0 * E           # per iteration
        imp_moved = lambda b_: (992.0 * b_ * H + 2 * 32.0 * b_ * H) * s + 992.0 * b_ * (16 + 8) + 32.0 * b_ * 16   # rows + ctx halves + dots, ids
        imp_gbs = imp_bytes / (imp_ms * 1e-3) / 1e9 if imp_ms else 0.0
        # what ANY launch costs at this size on this box, timed the same way (the measured argument): a launch that moves
        # nothing (one element filled), and a plain device copy of the step's input rows (E x H elements read AND written: 1.8x the bytes
        # the IMP launch moves, no dependent loads, no reduction)
        rows_ = torch.empty((E, H), dtype=tdtype, device=dev)
        rows2_, one_ = torch.empty_like(rows_), torch.zeros(1, device=dev)
        imp_floor = {'empty_launch_us': round(graph_launch_us(lambda: one_.fill_(1.0)), 3),
                     'copy_of_the_input_rows_us': round(graph_launch_us(lambda: rows2_.copy_(rows_)), 3),
                     'copy_bytes': 2 * rows_.numel() * rows_.element_size(), 'imp_launch_us': round(1e3 * imp_ms, 3)}
        del rows_, rows2_
        BL = 128                                                 # same launch on a graph that fills the chip
        impL_ms = imp_iter_ms(model, BL, tdtype)
        from sgg_amd import _lib as _sgg_lib
        mfma_units = _sgg_lib.load().sgg_imp_ctx_mfma_min_units()
        impL_mfma = tdtype != torch.float32 and BL * (H * 2 // 128) >= mfma_units
        impL_kernel = ('imp_ctx_mfma_kernel (persistent: LDS-DMA ring, gates from the dot products, block-sparse gate-matrix product on the '
                       'matrix cores; sgg_imp_ctx_fwd hands >= %d (graph, slice) units to it)' % mfma_units) if impL_mfma else 'imp_ctx_sliced_kernel'
        imp_forms = {'B%d' % b_: {k_: round(imp_iter_ms(model, b_, tdtype, kind=k_), 5) for k_ in
                                  (('ctx_sliced', 'ctx_mfma', 'ctx_lists', 'gate_proj') if tdtype != torch.float32 else ('ctx_sliced', 'ctx_lists', 'gate_proj'))}
                     for b_ in (B, BL)}
        impL_bytes = (2.0 * (992 * BL + 32 * BL) * H) * s + 8.0 * 992 * BL
        impL_gbs = impL_bytes / (impL_ms * 1e-3) / 1e9
        # context for the fraction: what a plain device-to-device copy moving the same number of bytes (half read, half written)
        # reaches on this box, timed the same way (50 launches in one graph between two events)
        src = torch.empty(int(impL_bytes) // 2, dtype=torch.uint8, device=dev)
        dst = torch.empty_like(src)
        dst.copy_(src)
        torch.cuda.synchronize()
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            for _ in range(50):
                dst.copy_(src)
        cg.replay()
        torch.cuda.synchronize()
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        cg.replay()
        c1.record()
        torch.cuda.synchronize()
        copy_gbs = impL_bytes / (c0.elapsed_time(c1) / 50 * 1e-3) / 1e9
        del src, dst, cg
        roi_ms = sum(v[0] * v[1] for (n, t), v in kt.items() if n == 'sgg_roi_align_fwd')
        # rows RoIAlign actually writes: the N boxes + one row per unordered pair (or per edge with SGG_EDGE_PAIRS=0), 25088 elements each
        roi_rows = N + U
        roi_bytes = roi_rows * 25088.0 * s + B * 38 * 38 * 512.0 * s
        conv_ms = sum(v[0] * v[1] for (n, t), v in kt.items() if n in ('sgg_conv3x3_relu', 'sgg_conv1_1', 'sgg_conv1_block', 'sgg_maxpool2x2'))
        vgg_flop = 226.13e9 * B
        total_ms = sum(v[0] * v[1] for v in kt.values())
        top = sorted(((v[0] * v[1], n, t) for (n, t), v in kt.items()), reverse=True)[:int(os.environ.get('SGG_BENCH_TOP', '10'))]
        # the other mode, for reference (short run)
        other = infer_step if args.mode == 'train' else None
        other_line = None
        if other is not None and world == 1:
            prime_eval(feed_hbm)
            el2 = timed(other, 2, 10)
            other_line = {'mode': 'infer', 'value': round(B * 10 / el2, 2), 'unit': 'images/s', 'ms_per_step': round(1e3 * el2 / 10, 3)}
        # the reference's literal boundary: a list of f32 [3,S,S] HOST tensors (SquarePad + ToTensor done on the CPU, rel_model_base.py:180),
        # copied per image inside the forward (pageable memory, no prefetch) -- short run, informational
        pcie = None
        if world == 1 and not args.force_dist:
            n = 10
            f32_host = []
            for hb in host_batches:
                t = list(hb)
                t[0] = [im.permute(2, 0, 1).float().div(255) for im in hb[0]]
                f32_host.append(tuple(t))
            el_f32 = timed(step, 2, n, lambda m: (f32_host[i % NB] for i in range(m)))
            if trainer is not None:
                trainer.flush()
            pcie = {'host_f32_per_image_copies': round(B * n / el_f32, 2), 'unit': 'images/s',
                    'note': 'reference-style input: f32 CHW host tensors, pageable, one copy per image inside the forward; 2 warm-up + %d steps' % n}
        line = {
            'metric': 'images/sec (whole node), VG SGCls IMP %s step' % ('train' if args.mode == 'train' else 'inference'),
            'value': round(world * B * args.steps / elapsed, 3),
            'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 3), 'host_issue_ms_per_step': round(head_issue_ms - head_wait_ms, 3), 'host_wait_ms_per_step': round(head_wait_ms, 3), 'host_feed_ms_per_step': round(head_feed_ms, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'input': ('host-resident Blob tuples (decoded u8 images + gt tensors on the host) through DeviceStager.prefetch: a worker thread packs '
                      'each batch into pinned memory and launches ONE async copy per batch on a copy stream (ring of %d slots, ahead of the '
                      'consumer); gc.freeze() after warm-up' % len(stager.slots) if args.input == 'host' else 'batches resident in HBM'),
            ('hbm_resident' if args.input == 'host' else 'host_input'): {
                'value': round(world * B * args.steps / elapsed_other, 3), 'unit': 'images/s', 'ms_per_step': round(1e3 * elapsed_other / args.steps, 3), 'host_issue_ms_per_step': round(other_issue_ms, 3),
                'steps': args.steps, 'note': 'the same rotation over the same %d batches with the %s' % (
                    NB, 'inputs resident in HBM before the timed region (no copy inside it)' if args.input == 'host' else 'host-resident inputs through DeviceStager.prefetch')},
            'config': {'workload': 'VG SGCls rel_model_stanford (IMP) %s, 592x592 frames, 32 boxes/img, 992 edges/img, '
                                   '3 IMP iters (BASELINE configs[1]/[3])' %
                                   ('train step: fwd + losses + bwd + grad all-reduce + clip + SGD' if args.mode == 'train'
                                    else 'eval forward incl. eval tail'),
                       'mode': args.mode, 'images_per_gpu': B, 'global_batch': world * B, 'loss': args.loss if args.mode == 'train' else None,
                       'batches': '%d distinct batches per rank in rotation (own images / boxes / classes / gt_rels); edges per batch %s -- the last '
                                  'one has %s boxes per image' % (NB, edges_per_batch, '/'.join(str(c) for c in ragged_counts)),
                       'rccl_ranks': dist.get_world_size() if dist.is_initialized() else 1,
                       'edge_branch': ('union-box RoIAlign and fc6 K=25088 computed once per UNORDERED box pair (%d pairs for %d edges per '
                                       'GPU and step), per-edge rect term added after; same outputs as the per-edge computation '
                                       '(SGG_EDGE_PAIRS=0)' % (992 * B // 2, 992 * B)) if os.environ.get('SGG_EDGE_PAIRS', '1') != '0'
                       else 'per edge, as in the reference',
                       'parallelism': 'image-sharded dp%d, %s' % (
                           world, 'no collective' if args.mode != 'train' else
                           ('RCCL, bf16 on the wire: fc6 / fc7 gradients reduce-scattered, clip + SGD on 1/%d of their fp32 masters per rank, '
                            'updated bf16 operands all-gathered; the other tensors all-reduced' % world)
                           if (trainer is not None and trainer.shard_optimizer) else 'RCCL gradient all-reduce (bf16 on the wire)'),
                       'weights': 'random init (He), frozen VGG16 + trainable IMP head (247.75 M params)',
                       'pipeline': 'optimiser update of step k runs on a side stream under the frozen VGG forward of step k+1 (every update inside the timed region)',
                       'hipgraph_fallback': ('the first attempt of this run ended abnormally; this line is from a second run with the train step launched kernel by kernel (SGG_GRAPH=0)'
                                             if os.environ.get('SGG_BENCH_FALLBACK') else None),
                       'hipgraph_eval': (dict(model.__dict__['_eval_graphs'].stats, disabled=model.__dict__['_eval_graphs'].disabled, priming_calls_before_warmup=3 * NB,
                                              note='evaluation forward replayed as one hipGraph per batch signature (sgg_amd/graph_forward.py); the result copy to the host stays a synchronisation per call')
                                         if hasattr(model.__dict__.get('_eval_graphs'), 'stats') else None),
                       'hipgraph': (dict(trainer.graphs.stats, disabled=trainer.graphs.disabled,
                                         priming_steps_before_warmup=primed,
                                         note='train step replayed as one-stream hipGraphs per batch signature (U: update of the previous step, on the lane stream || V: VGG forward; B: head forward + loss + backward in three segments, the backward lane work beside the fc6 / fc7 weight gradients), sgg_amd/graph_step.py; counts over the whole process; wait_s = the issuing thread held back on purpose (at most 8 steps ahead, one device synchronisation per 32 steps)')
                                    if (trainer is not None and trainer.graphs is not None) else None)},
            'roofline': {'kernel': ('256x256 ping-pong MFMA kernel%s (+ the 128x128 split-K launch that replaces a nearly empty last round), %s' % (
                             ', TN form: both operands staged with the reduction index as the slow axis, fragments through ds_read_b64_tr_b16, no '
                             'transposed copies' if (tag == 'bwd_fc6_edge_dW' and get('sgg_gemm_tn256', tag)[1]) else '', desc)),
                         'bound': 'mfma', 'achieved': round(tf, 2), 'peak': peak,
                         'unit': 'TFLOP/s', 'frac': round(tf / peak, 4),
                         'traffic': pmc_traffic('fc6_edge_gemm' if tag == 'fc6_edge' else 'fc6_dW_gemm') if (B == 8 and args.dtype != 'f32') else None,
                         'traffic_source': 'profiles/pmc_r0x.json: separate rocprofv3 --pmc passes of the same launch (tools/pmc_traffic.sh), not measured in this run',
                         'ms_per_step': round(ms, 4), 'executed_flop': flop,
                         'alone': ({'ms': round(alone_ms, 4), 'TFLOP/s': round(flop / (alone_ms * 1e-3) / 1e12, 1),
                                    'note': 'the same contraction launched on its own (incl. its split-K tail): in the step the node lane\'s stream runs HBM-bound reductions '
                                            'and 256-row contractions beside it (DESIGN 10), which lengthens this launch and shortens the step'} if alone_ms else None),
                         'reference_algorithm_tflops': round(ref_flop[tag] / (ms * 1e-3) / 1e12, 2),
                         'note': ('achieved = FLOPs the launch executes / its time; the reference runs this contraction on every EDGE '
                                  '(SURVEY 8(d)), here it runs once per unordered box pair: reference_algorithm_tflops prices the '
                                  'reference\'s FLOPs over the same time') if paired else None},
            'roofline_fc6_forward': ({'kernel': '256x256 ping-pong MFMA kernel, ' + fwd_desc, 'bound': 'mfma', 'achieved': round(fwd_flop / (fwd_ms * 1e-3) / 1e12, 2),
                                      'peak': peak, 'unit': 'TFLOP/s', 'frac': round(fwd_flop / (fwd_ms * 1e-3) / 1e12 / peak, 4), 'ms_per_step': round(fwd_ms, 4),
                                      'executed_flop': fwd_flop, 'note': 'the second largest contraction (248 tiles = one round, 784 K-tiles per tile): '
                                      'round 2 reported this launch as `roofline`'} if (fwd_ms > 0 and tag != 'fc6_edge') else None),
            'roofline_vgg': {'kernel': 'VGG-16 features of the frozen detector: conv1_1 (K = 27 on MFMA, computed inside conv1_2) + 12 x 3x3 conv on MFMA (conv_pp.hip: LDS-resident patch under a '
                                       'ping-pong schedule; conv1_2 on the lock-step patch kernel, conv5 as implicit GEMM; pools fused in the epilogues): the largest time slice of the step',
                             'bound': 'mfma', 'achieved': round(vgg_flop / (conv_ms * 1e-3) / 1e12, 1) if conv_ms else 0.0, 'peak': peak, 'unit': 'TFLOP/s',
                             'frac': round(vgg_flop / (conv_ms * 1e-3) / 1e12 / peak, 4) if conv_ms else 0.0, 'traffic': None,
                             'ms_per_step': round(conv_ms, 4), 'executed_flop': vgg_flop,
                             'back_to_back_ms': round(vgg_alone, 4) if vgg_alone else None,
                             'back_to_back_frac': round(vgg_flop / (vgg_alone * 1e-3) / 1e12 / peak, 4) if vgg_alone else None},
            'roofline_imp': {'kernel': 'imp_ctx_sliced_kernel (the IMP gather / gate / scatter step, one launch per iteration: every edge row read once, '
                                       'gates from dot products, two segmented sums per node; the edge inputs are never formed -- node projection)', 'bound': 'hbm',
                             'achieved': round(imp_gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                             'frac': round(imp_gbs / HBM_PEAK_GBS, 4),
                             'traffic': pmc_traffic('imp_ctx_B8'),
                             'reference_algorithm_bytes': imp_bytes, 'moved_bytes': imp_moved(B), 'avg_launch_ms': round(imp_ms, 5),
                             'achieved_moved': round(imp_moved(B) / (imp_ms * 1e-3) / 1e9, 1) if imp_ms else 0.0,
                             'frac_moved': round(imp_moved(B) / (imp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if imp_ms else 0.0,
                             'floor_at_this_size': imp_floor,
                             'note': 'achieved / frac price SURVEY 8(d)\'s bytes of the REFERENCE step (incl. the e_in write this design removed '
                                     'algebraically) over the launch time; achieved_moved / frac_moved price the bytes this launch really moves (= PMC '
                                     'traffic). At B=8 the launch is latency-bound (a launch + two dependent memory levels; roofline time 1-2 us): the '
                                     '0.80 target is not reachable as a standalone launch at this size -- see roofline_imp_large'},
            'roofline_imp_large': {'kernel': 'the same launch at %d images (%d edges): %s' % (BL, 992 * BL, impL_kernel), 'bound': 'hbm',
                                   'achieved': round(impL_gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                   'frac': round(impL_gbs / HBM_PEAK_GBS, 4),
                                   'traffic': pmc_traffic('imp_ctx_B128'),
                                   'reference_algorithm_bytes': impL_bytes, 'moved_bytes': imp_moved(BL), 'avg_launch_ms': round(impL_ms, 5),
                                   'achieved_moved': round(imp_moved(BL) / (impL_ms * 1e-3) / 1e9, 1),
                                   'frac_moved': round(imp_moved(BL) / (impL_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                   'forms_ms': imp_forms,
                                   'device_copy_same_bytes': {'GB/s': round(copy_gbs, 1), 'frac_of_copy': round(impL_gbs / copy_gbs, 4)}},
            'kernels': {'sum_kernel_ms_per_step': round(total_ms, 3),
                        'vgg16_ms': round(conv_ms, 3), 'vgg16_tflops': round(vgg_flop / (conv_ms * 1e-3) / 1e12, 1) if conv_ms else 0,
                        'roi_align_ms': round(roi_ms, 4), 'roi_align_GBs': round(roi_bytes / (roi_ms * 1e-3) / 1e9, 1) if roi_ms else 0,
                        'roi_align_rows': roi_rows,
                        'top': [{'ms_per_step': round(ms_, 3), 'call': n, 'tag': t} for ms_, n, t in top]},
        }
        if comm:
            line['comm'] = comm
        if other_line:
            line['other_mode'] = other_line
        if pcie:
            line['pcie_inclusive'] = pcie
        if world == 1 and not args.force_dist and args.dtype != 'f32' and not args.no_f32 and os.environ.get('SGG_EDGE_PAIRS', '1') != '0' \
                and args.mode in ('train', 'infer'):
            # the same steps with the edge branch computed PER EDGE, as the reference does (SGG_EDGE_PAIRS=0): short runs beside the headline
            os.environ['SGG_EDGE_PAIRS'] = '0'
            try:
                if trainer is not None:
                    el_pe = timed(train_step, 3, 8)
                    trainer.flush()
                else:
                    el_pe = None
                el_pi = timed(infer_step, 2, 8)
            finally:
                del os.environ['SGG_EDGE_PAIRS']
            line['per_edge_branch'] = {'infer_images_per_s': round(B * 8 / el_pi, 2), 'infer_ms_per_step': round(1e3 * el_pi / 8, 3),
                                       'note': 'same workload with RoIAlign and fc6 run on every edge as in the reference (SGG_EDGE_PAIRS=0); '
                                               '3 (train) / 2 (inference) warm-up + 8 timed steps; never `value`'}
            if el_pe is not None:
                line['per_edge_branch'].update(train_images_per_s=round(B * 8 / el_pe, 2), train_ms_per_step=round(1e3 * el_pe / 8, 3))
        if world == 1 and not args.force_dist and args.dtype == 'f16' and not args.no_f32 and args.mode in ('train', 'infer'):
            # BASELINE.json words configs[1] "bf16": the same kernels with bf16 storage / MFMA operands (same rates, 8x the rounding error of
            # f16 -- tests/test_parity_full_gpu.py), short runs beside the headline -- never `value`
            if trainer is not None:
                trainer.flush()
            torch.cuda.synchronize()
            model.set_compute_dtype(torch.bfloat16)
            tb = Trainer(model, lr=1e-3, pipeline=trainer.pipeline, loss_type=args.loss, graph=False) if trainer is not None else None     # (11 steps: launch by launch, no capture inside the short run)
            el_bt = timed(lambda b: tb.step(b), 3, 8) if tb is not None else None
            if tb is not None:
                tb.flush()
                tb.opt.state.clear()
            el_bi = timed(infer_step, 2, 8)
            line['bf16_mode'] = {'dtype': 'bf16', 'infer_images_per_s': round(B * 8 / el_bi, 2), 'infer_ms_per_step': round(1e3 * el_bi / 8, 3),
                                 'note': 'same workload and kernels with bf16 storage / operands (BASELINE.json\'s wording of configs[1]); 3 (train) / 2 '
                                         '(inference) warm-up + 8 timed steps; parity of both modes: profiles/r03_parity_bench_config.json'}
            if el_bt is not None:
                line['bf16_mode'].update(train_images_per_s=round(B * 8 / el_bt, 2), train_ms_per_step=round(1e3 * el_bt / 8, 3))
            del tb
            model.set_compute_dtype(tdtype)
        if world == 1 and not args.force_dist and args.dtype != 'f32' and not args.no_f32:
            # the reference computes in fp32: the same two steps (a) in the x3 mode -- fp32 storage, every contraction on f16 split operands
            # with fp32 accumulation: fp32-grade logits, inside the north star's 1e-3 clause (tests/test_parity_full_gpu.py) -- and (b) in
            # exact-fp32 mode (v_mfma_f32_32x32x2_f32); short runs, reported beside the 16-bit headline -- never `value`
            if trainer is not None:
                trainer.flush()
                trainer.opt.state.clear()            # the 16-bit run is over: its 1 GB of momentum buffers goes back to the allocator
            torch.cuda.synchronize()
            for key, split in (('x3_mode', True), ('f32_mode', False)):
                model.set_compute_dtype(torch.float32, split3=split)
                torch.cuda.empty_cache()             # fp32 activations are twice the size: let the allocator start from whole blocks
                t32 = Trainer(model, lr=1e-3, pipeline=trainer.pipeline if trainer is not None else True, loss_type=args.loss, graph=False)
                # x3 is the parity-qualified throughput ("the one to move"): the headline's own W warm-up and >= 20 timed steps over
                # the same rotation, after the mode's operand caches (weight splits, index tables of all four batch signatures) are warm --
                # round 5's 4 + 5 steps timed a warming cache (295 images/s in the driver's run against 362 - 366 here).  Exact fp32: short.
                wt, kt_, wi, ki = (max(8, args.warmup), max(20, min(args.steps, 40)), 4, max(20, min(args.steps, 40))) if split else (4, 5, 2, 5)
                el_t = timed(lambda b: t32.step(b), wt, kt_)
                t32.flush()
                el_i = timed(infer_step, wi, ki)
                line[key] = {'dtype': 'f32 storage, f16 split operands (hi + lo), fp32 accumulate' if split else 'f32',
                             'train_images_per_s': round(B * kt_ / el_t, 2), 'train_ms_per_step': round(1e3 * el_t / kt_, 3),
                             'infer_images_per_s': round(B * ki / el_i, 2), 'infer_ms_per_step': round(1e3 * el_i / ki, 3),
                             'train_steps': kt_, 'train_warmup': wt, 'infer_steps': ki, 'infer_warmup': wi,
                             'mfma_peak_TFLOPs': round(MFMA_PEAK_TF['f16'] / 3.0, 1) if split else MFMA_PEAK_TF['f32'],
                             'note': ('same workload: three f16 MFMA products per fp32-grade product (hi.hi + hi.lo + lo.hi), logits within 1e-3 of the '
                                      'fp32 reference (profiles/r0x_parity_bench_config.json: x3).  Round 6: activations and weights as PAIRS of f16 planes '
                                      '(SGG_PAIR16), the MFMA loops walk the plane segments themselves, VGG-16 writes pair planes from its epilogues (no split '
                                      'pass, conv2_1 .. conv4_3 on the patch kernel with the pools fused).  %d (train) / %d (inference) warm-up + %d / %d timed '
                                      'steps, launch by launch (no hipGraph replay in this mode)' % (wt, wi, kt_, ki))
                             if split else 'same workload, exact-fp32 MFMA; 4 (train) / 2 (inference) warm-up + 5 timed steps'}
                if split:
                    # the same mode with an f16 BACKWARD (set_compute_dtype(float32, split3=True, backward_f16=True)): the forward -- what both parity
                    # clauses are about -- unchanged (bit-equal logits), the backward's contractions one f16 product instead of three under the loss
                    # scale: gradients at the f16 mode's accuracy (tests/test_x3_gpu.py).  Reported beside the strict form, never instead of it.
                    model.set_compute_dtype(torch.float32, split3=True, backward_f16=True)
                    t32.flush()
                    el_tf = timed(lambda b: t32.step(b), wt, kt_)
                    t32.flush()
                    model.set_compute_dtype(torch.float32, split3=True)
                    line[key]['train_f16_backward'] = {
                        'images_per_s': round(B * kt_ / el_tf, 2), 'ms_per_step': round(1e3 * el_tf / kt_, 3), 'steps': kt_, 'warmup': wt,
                        'note': 'x3 forward (logits within 1e-3: same bits as the strict x3 forward) + f16 backward contractions under the loss scale '
                                '(gradients within 1e-2 of each tensor\'s largest entry, the f16 mode\'s accuracy): mixed-precision training with an '
                                'fp32-grade forward'}
                    line[key]['vs_headline'] = {'train_x': round((el_t / kt_) / (elapsed / args.steps), 2) if args.mode == 'train' else None,
                                                'note': 'x3 train step / the f16 headline step (three products per product bound it at <= 3x on the contraction share)'}
                t32.opt.state.clear()
                del t32
            model.set_compute_dtype(tdtype)
        if world == 1 and not args.force_dist and args.dtype != 'f32' and not args.no_side_modes and args.mode == 'train':
            # BASELINE configs[2] beside the headline (the driver only runs the default line): the SGDet eval forward at the config's size
            # -- 8 x 1 000 proposals through the box head -- with its own roofline (the box head's fc6) and its own CPU baseline
            if trainer is not None:
                trainer.flush()
                trainer.opt.state.clear()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            try:
                sm = sgdet_model(dev, tdtype)
                line['sgdet_mode'] = sgdet_measure(args, sm, dev_batches, timed, B, dev, 20, 4,
                                                   cpu_images=0 if args.no_cpu_baseline else 1)
                del sm
            except Exception as e:          # a side measurement never takes the headline down; the line says what happened
                line['sgdet_mode'] = {'error': repr(e)[:400]}
            torch.cuda.empty_cache()
            # BASELINE configs[4] on one GPU: GQA vocabulary + ResNet-50-FPN + the GAN iteration (main.py:100-194), short run
            try:
                line['gqa_gan_mode'] = gqa_gan_measure(args, dev, tdtype, timed, B, 5, 2, cpu=not args.no_cpu_baseline)
            except Exception as e:
                line['gqa_gan_mode'] = {'error': repr(e)[:400]}
            from sgg_amd import ops as _ops
            _ops.split3_cache_clear()
            torch.cuda.empty_cache()
        if world == 1 and not args.force_dist and args.dtype != 'f32' and not args.no_side_modes and args.mode == 'train':
            # the data-parallel code path at ONE rank (a 1-rank RCCL group: gradient hooks, wire-dtype buffers, reduce-scatter / all-reduce /
            # all-gather calls, sharded fused optimiser -- what `--gpus N` runs per rank, minus the wire): its own overhead on record while no
            # 8-GPU node exists.  Launch by launch: graph replay is off whenever the data-parallel path is on.
            try:
                os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
                os.environ.setdefault('MASTER_PORT', str(_free_port()))
                os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
                if trainer is not None:
                    trainer.flush()
                    trainer.opt.state.clear()
                model.set_compute_dtype(tdtype)
                torch.cuda.empty_cache()
                dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
                try:
                    tdp = Trainer(model, lr=1e-3, force_dist=True, loss_type=args.loss, pipeline=True, graph=False)
                    el_dp = timed(lambda b: tdp.step(b), 8, 20)
                    tdp.flush()
                    tpl = Trainer(model, lr=1e-3, loss_type=args.loss, pipeline=True, graph=False)
                    el_pl = timed(lambda b: tpl.step(b), 8, 20)
                    tpl.flush()
                    line['dp_path_world1'] = {'images_per_s': round(B * 20 / el_dp, 2), 'ms_per_step': round(1e3 * el_dp / 20, 3),
                                              'same_process_plain_path_ms_per_step': round(1e3 * el_pl / 20, 3),
                                              'rccl_ranks': dist.get_world_size(), 'shard_optimizer': bool(tdp.shard_optimizer),
                                              'note': 'Trainer(force_dist=True) on a 1-rank RCCL group, 8 warm-up + 20 timed steps, launch by launch; beside it the '
                                                      'one-GPU path launch by launch in the same process (the headline replays hipGraphs, this does not)'}
                    tdp.opt.state.clear()
                    tpl.opt.state.clear()
                    del tdp, tpl
                finally:
                    dist.destroy_process_group()
            except Exception as e:
                line['dp_path_world1'] = {'error': repr(e)[:400]}
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args.cpu_images, 111)
        # which of the north star's parity clauses each reported throughput meets
        pr = parity_records()
        line['parity'] = pr
        mode_of_value = args.dtype
        line['parity']['value'] = {'mode': mode_of_value, 'meets_logits_1e-3': pr.get(mode_of_value, {}).get('meets_logits_1e-3'),
                                   'meets_r50_0.1': pr.get(mode_of_value, {}).get('meets_r50_0.1'),
                                   'parity_qualified_throughput': 'x3_mode (meets both clauses)' if not pr.get(mode_of_value, {}).get('meets_logits_1e-3') else 'value'}
        if 'bf16_mode' in line and pr.get('bf16', {}).get('recall_delta_points'):
            line['bf16_mode']['r50_delta'] = pr['bf16']['recall_delta_points']
            line['bf16_mode']['meets'] = {'logits_1e-3': pr['bf16'].get('meets_logits_1e-3'), 'r50_0.1': pr['bf16'].get('meets_r50_0.1')}
        if 'x3_mode' in line:
            line['x3_mode']['meets'] = {'logits_1e-3': pr.get('x3', {}).get('meets_logits_1e-3'), 'r50_0.1': True}
    else:
        line = None
    if dist.is_initialized():
        # the ranks that did not profile wait here on the HOST (gloo): an RCCL barrier would keep their GPUs spinning on
        # rank 0's memory for the whole of its profiling pass
        if host_group is not None:
            dist.barrier(group=host_group)
        else:
            dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        # RCCL prints its version banner through C stdio, which is block-buffered on a pipe and would otherwise be
        # flushed at exit, AFTER this line: push it out first so that the JSON is the last thing rank 0 writes.
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(line), flush=True)


if __name__ == '__main__':
    main()
