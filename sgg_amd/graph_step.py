"""The train step of sgg_amd.trainer.Trainer as replayed hipGraphs.

The launch-by-launch step issues ~210 kernels from Python (81 in the head's forward, 89 in its backward, the VGG-16 forward, the fused
optimiser): 3.5 - 5.5 ms of the issuing thread per 6.9 ms step -- issue-bound on a loaded host.  Here a step is THREE graph launches:

    U  (node lane's stream)   clip + SGD on the PREVIOUS step's gradients, rebuild of the weight-derived operands
    V  (calling stream)       VGG-16 forward of this step's images                                   -- U || V, joined by one event --
    B  (calling stream)       RoIAlign, the head's forward, the loss, the head's backward            (gradients left in B's own static tensors)

i.e. the pipelined step of Trainer (the update of step k under the frozen VGG forward of step k + 1).  Every graph is a ONE-STREAM graph:
on this runtime (ROCm 7.0 HIP, tools/graph_probe.py) a graph whose nodes span two streams costs its launch 2.6 ms of the calling thread
and back-to-back launches of such graphs stall each other, a one-stream graph of the same kernels launches in 0.07 ms -- so the streams
meet BETWEEN graphs (plain events), and the lanes the launch-by-launch path forks inside the head are off during a capture.
Everything a replay reads lives at one address: the batch is copied into static input tensors (one multi-tensor copy), the dropout seed
of the step is written into device memory (sgg_dropout_fwd_dev reads it), the feature map is the detector's `out=` tensor, the
weight-derived operands are rewritten in place by U (shadows, transposed copies, W6's group sums).  What a graph bakes in by value --
learning rates, loss scale, the row counts of the batch's relation labels -- is part of its key.

    B is captured per batch SIGNATURE (image sizes, boxes per image, relation rows and what the host knows about them), V per B;
    U per (which B ran before it: that B's tensors hold the gradients the update reads; the optimiser's scalars).

A signature's first steps run launch by launch (they fill the per-signature caches the capture must not touch: index tables, segment
tensors); then B is captured and replayed, and from the next step of that kind on U and V.  Only the process that owns the GPU captures, on
its own streams; anything the capture does not take (data-parallel steps, fp32 modes, sub-sampled relation lists, host-resident inputs, a
model that is not RelModelStanford in a gt-box mode) -- and any error during a capture -- leaves the plain launch-by-launch path of
Trainer.step, never another process.  SGG_GRAPH=0 turns the whole thing off.

Bit-equality: a replayed step launches the kernels of the launch-by-launch step with the same arguments (the lanes are scheduling only,
never a different sum), so parameters after n graph steps equal those after n plain steps bit for bit (tests/test_graph_gpu.py)."""
import os
import warnings

import torch

from . import _lib, ops
from .rel_model_base import host_of, image_hw, rels_host_facts

WARM_STEPS = 3        # launch-by-launch steps of a new signature before its capture


class _B(object):
    """one captured head step: graph, its static inputs, the tensors it leaves its results in"""
    __slots__ = ('segments', 'static', 'loss', 'pgrads', 'local', 'fmap', 'sizes', 'padded', 'sig', 'replays', 'pins')


def pin_caches(model):
    """References to every per-signature index tensor a capture may have baked the address of: the model's candidate-edge
    cache (rel_inds, CSR lists, pairing -- rel_model_stanford.py `_graph_cache`), its segment tensors (rel_model_base.py `_seg_cache`)
    and the unordered-pair tables (pairing._TABLES).  Those caches evict by clearing themselves; a graph that holds the entries it was
    captured with keeps their storage alive (and at its address) for as long as the graph lives -- the caches may forget them, the
    plain path then simply builds equal tensors elsewhere.  Small: index tensors of a few MB per signature."""
    from . import pairing
    if os.environ.get('SGG_GRAPH_NO_PINS') == '1':        # (tests only: shows that the eviction test fails without the pins)
        return None
    return (dict(model.__dict__.get('_graph_cache') or {}), dict(model.__dict__.get('_seg_cache') or {}), dict(pairing._TABLES))


def _dbg(tag):
    """SGG_GRAPH_DEBUG=1: synchronise after every graph launch and say which one it was (a faulting replay then names itself)"""
    if os.environ.get('SGG_GRAPH_DEBUG') == '1':
        import sys
        torch.cuda.synchronize()
        sys.stderr.write('[graph] %s ok\n' % tag)
        sys.stderr.flush()


class _Seq(object):
    """graphs replayed back to back on the current stream (graph V in one or two parts)"""

    def __init__(self, parts):
        self.parts = list(parts)

    def replay(self):
        for g in self.parts:
            g.replay()


class GraphStep(object):
    def __init__(self, trainer):
        import weakref
        self.tr = weakref.proxy(trainer)          # (no cycle trainer <-> graphs: the captured pools go when the trainer goes)
        self.model = trainer.model
        self.warm = {}            # signature -> launch-by-launch steps seen
        self.B = {}               # signature -> _B
        self.U = {}               # (id of the B whose gradients it applies, optimiser scalars) -> graph: clip + SGD + rebuild of the derived operands
        self.V = {}               # (id of the B it feeds) -> graph: VGG-16 forward of B's static images into B's feature map
        self.pending = None       # the _B whose gradients have not been applied yet
        self._last_pending = None # ... as it was when the current step() began (what an exception handler needs to know)
        self.pending_scalars = None   # ... and the optimiser's scalars at the end of that step (the plain path queues the update there)
        self.generation = None
        self.disabled = None      # reason (str) once a capture failed
        self.pool = None
        self.seed = None
        self.stats = dict(replayed=0, plain=0, captures=0, wait_s=0.0)
        # run-ahead bound: the calling thread issues a replayed step in ~0.4 ms, the GPU needs ~7 -- with nothing else to stop it the host queues
        # hundreds of graph launches, and on this runtime (ROCm 7.0 HIP) that ends in "Memory access fault by GPU": 200 unsynchronised steps
        # (1000 launches in flight) faulted in 12 of 12 bench runs, 25 and 60 steps in none (tools/graph_fault_stats.sh).  The host waits for
        # the step issued DEPTH steps ago before it issues the next one: the GPU's queue never runs dry, the runtime's never overflows.
        self.max_signatures = int(os.environ.get('SGG_GRAPH_SIGNATURES', '8'))
        self.depth = max(1, int(os.environ.get('SGG_GRAPH_DEPTH', '8')))
        self.inflight = []
        # ... and that alone was not enough: with the host held to 2 or 8 steps ahead by EVENT waits a 400-step run still faulted, with a device-wide
        # synchronisation every 32 steps it does not (5 of 5 bench runs of 200 / 400 steps, tools/graph_probe.py 400).  Cost: the queue drains
        # once per 32 steps (~0.4 ms of 220).
        # W^T copies of the backward made on the lane inside B's forward instead of inside U: measured (round 5) U alone 1.65 -> 1.56 ms, B 4.18 ->
        # 4.27 ms, the step 6.78 -> 6.86 ms -- not kept (off), the switch stays for a box where the update is the longer half of the pair
        self.defer_transposes = os.environ.get('SGG_GRAPH_DEFER_TRANSPOSES', '0') == '1'
        # the rebuild of the derived operands as a graph of its own on the calling stream after the pair has met (its dozen short launches then
        # pay no kernel-boundary penalty beside the forward): measured 6.92 - 7.08 ms per step against 6.81 - 6.88 with the rebuild inside U -- off
        self.split_update = os.environ.get('SGG_GRAPH_SPLIT_UPDATE', '0') == '1'
        # Round 6: the fault does not come back with today's tree -- 10 of 10 bench runs of 400 replayed steps, one of 1500 and two with the host
        # 64 steps ahead end clean WITHOUT the periodic synchronisation (profiles/r06_graph_replay.txt; even with the empty graphs kept), and
        # neither a HIP-only nor a torch-only program with the same launch pattern ever faulted (tools/native/graph_replay.hip,
        # tools/graph_replay_torch.py).  What round 5 met was fixed by one of its own later changes (no empty graphs, no captured memset
        # nodes) and the synchronisation had stayed as a precaution: off by default now (SGG_GRAPH_SYNC_EVERY=n brings it back), the
        # event-based run-ahead bound stays (it costs nothing: the host waits for a step that finished long ago).
        self.sync_every = int(os.environ.get('SGG_GRAPH_SYNC_EVERY', '0'))
        self.since_sync = 0
        # graph V in two parts, the update released after the first: the forward's first layers (the fused conv1 block, conv2_x: large maps, short
        # reductions) are the ones that suffer beside the update's 5 GB stream (kernel trace of a replayed step: conv1 block 281 -> 425 us,
        # conv2_1 / conv2_2 168 -> 251 / 296 us, conv3_1 ... conv5_3 unchanged) -- so they run first, alone, and the update runs beside the rest.
        # The number = convolutions in the first part (0: one graph, the update beside all of it)
        self.v_split = int(os.environ.get('SGG_GRAPH_VSPLIT', '0'))

    # ------------------------------------------------------------------ what the capture takes
    def _signature(self, batch):
        """None when this batch has to go launch by launch; else a hashable description of everything the captured launches bake in"""
        tr, m = self.tr, self.model
        if self.disabled or tr.dist_on or not tr.pipeline or not ops.is_half(m.compute_dtype) or m.mode == 'sgdet' or m.backbone != 'vgg16':
            return None
        if getattr(m, 'use_bias', False) or not m.training:
            return None
        imgs, boxes, classes, rels = batch[0], batch[3], batch[4], batch[5]
        if rels is None or not all(torch.is_tensor(t) and t.is_cuda for t in (boxes, classes, rels)):
            return None
        if not all(torch.is_tensor(im) and im.is_cuda for im in imgs):
            return None
        ch, rh = getattr(classes, '_sgg_host', None), getattr(rels, '_sgg_host', None)
        if ch is None or rh is None or tuple(ch.shape) != tuple(classes.shape) or tuple(rh.shape) != tuple(rels.shape):
            return None
        im_ids = ch[:, 0].tolist()
        counts, order = {}, []
        for i in im_ids:
            if i not in counts:
                order.append(i)
                counts[i] = 0
            counts[i] += 1
        if order != list(range(len(imgs))):
            return None                                    # (images are indexed by their id: every image present, in order)
        # images are not resized (the transform's scale is 1 at the configuration: no per-box scale tensor from the host inside a capture)
        img_sig = tuple((tuple(im.shape), str(im.dtype)) for im in imgs)
        for im in imgs:
            h, w = image_hw(im)
            if m.detector.transform.resized_hw(h, w) != (h, w):
                return None
        facts = rels_host_facts(rh.tolist(), counts)
        R = int(rels.shape[0])
        n_cand = sum(c * (c - 1) for c in counts.values())
        if not facts['regular'] or R != facts['fg_pairs']:
            return None                                    # (duplicate relations on a pair: extra label rows, no cached tables)
        num_im = len(order)
        if min(R, int(m.RELS_PER_IMG * 0.25 * num_im)) < R or int(m.RELS_PER_IMG * num_im) - R < n_cand - R:
            return None                                    # the relation rows would be sub-sampled (a host-side random choice per step)
        return (img_sig, tuple(counts[i] for i in order), tuple(boxes.shape), str(boxes.dtype), R, facts['max_edges'], facts['max_per_pair'],
                str(m.compute_dtype), tr.loss_type, tr.loss_weights, float(tr.loss_scale), bool(tr.fused_loss), float(m.dropout_p),
                os.environ.get('SGG_EDGE_PAIRS', '1'))

    def _opt_scalars(self):
        o = self.tr.opt
        return (tuple(float(g['lr']) for g in o.param_groups), tuple(float(g['momentum']) for g in o.param_groups),
                tuple(float(g['weight_decay']) for g in o.param_groups), float(o.clip or 0.0), float(self.tr.loss_scale), int(o.max_blocks))

    # ------------------------------------------------------------------ the step
    def step(self, batch):
        """-> the loss of a replayed step, or None: the caller runs this batch launch by launch (after flush() has been seen to here)"""
        m = self.model
        gen = (getattr(m, '_operand_generation', 0), id(m))
        if self.generation != gen:
            self._drop('compute dtype changed')
            self.generation = gen
        if self.pending is not None and self._opt_scalars() != self.pending_scalars:
            # the learning rate (a scheduler's milestone) / the loss scale moved since the pending step ran: its update -- which the plain
            # path queued at the END of that step -- is applied now, with the values of then
            self._flush_with(self.pending_scalars)
        sig = self._signature(batch)
        if sig is None:
            self.flush()
            self.stats['plain'] += 1
            return None
        b = self.B.get(sig)
        self._last_pending = self.pending
        if b is None:
            # real data rarely repeats a signature (boxes per image vary): the bookkeeping stays bounded, and at most MAX_SIGNATURES kinds of
            # batch are ever captured (each holds its step's activations, ~0.5 GB per image at the bench size) -- the rest run launch by launch
            n = self.warm.get(sig, 0)
            if n < WARM_STEPS or len(self.B) >= self.max_signatures:
                if len(self.warm) > 4096:
                    self.warm.clear()
                self.warm[sig] = n + 1
                self.flush()
                self.stats['plain'] += 1
                return None
        if self.seed is None:
            self.seed = torch.zeros(1, dtype=torch.int64, device=batch[3].device)
        captured = b is not None
        try:
            import time
            t_w = time.perf_counter()
            while len(self.inflight) >= self.depth:
                self.inflight.pop(0).synchronize()
            self.since_sync += 1
            if self.sync_every and self.since_sync >= self.sync_every:
                torch.cuda.synchronize(self.seed.device)
                self.since_sync = 0
            self.stats['wait_s'] += time.perf_counter() - t_w       # the calling thread waiting for the GPU (not issuing): the run-ahead bound
            if b is None:
                b = self._capture_B(sig, batch)
            else:
                self._feed(b, batch)
                self._run_A(b)
            self._replay_B(b)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.seed.device))
            self.inflight.append(ev)
        except Exception as e:          # a capture that fails leaves the plain path, for good (and says why, once)
            if captured and self.U_ready(b):
                # nothing was being captured: an error out of a plain replay (out of memory, a bad batch, a fault) is the caller's to see
                self._abort_capture()
                raise
            if self.disabled is None:
                self.disabled = '%s: %s' % (type(e).__name__, e)
                warnings.warn('sgg_amd: hipGraph capture of the train step failed (%s); continuing launch by launch' % self.disabled, RuntimeWarning)
            self._abort_capture()
            self._drop('capture failed')            # (a pending update is applied launch by launch: nothing a capture records has run)
            if os.environ.get('SGG_GRAPH_STRICT') == '1':
                raise
            self.stats['plain'] += 1
            return None
        self.stats['replayed'] += 1
        return b.loss.detach().clone()

    def U_ready(self, b):
        """True when a step on `b` replays captured graphs only (B, its V, and the U of whatever is pending): no capture can be the source
        of an exception then"""
        m = self.model
        if (id(b), str(m.compute_dtype), self.v_split) not in self.V:
            return False
        prev = self._last_pending
        return prev is None or (id(prev), self._opt_scalars(), str(m.compute_dtype)) in self.U

    def flush(self):
        """apply the update a replayed step has left pending, launch by launch, on the current stream -- with the optimiser's scalars as they
        were at the END of that step (where the plain pipelined path queues its update): a scheduler that stepped between step() and flush()
        does not change the last update of an epoch"""
        b = self.pending
        if b is None:
            return
        if self.pending_scalars is not None and self._opt_scalars()[:3] != self.pending_scalars[:3] and not self._in_flush_with:
            return self._flush_with(self.pending_scalars)
        self.pending = None
        tr = self.tr
        self._point_grads(b)
        tr.opt.step(grad_scale=1.0 / tr.loss_scale, grads=dict(b.local), shards=None)
        tr._local = {}
        tr._queued = False
        from .train import train_weights
        train_weights(self.model)       # the derived operands too (in place): a replayed B that follows reads them without asking

    _in_flush_with = False

    def _flush_with(self, scalars):
        groups = self.tr.opt.param_groups
        self._in_flush_with = True
        now = [(g['lr'], g['momentum'], g['weight_decay']) for g in groups]
        for g, lr, mo, wd in zip(groups, scalars[0], scalars[1], scalars[2]):
            g['lr'], g['momentum'], g['weight_decay'] = lr, mo, wd
        try:
            self.flush()
        finally:
            self._in_flush_with = False
            for g, (lr, mo, wd) in zip(groups, now):
                g['lr'], g['momentum'], g['weight_decay'] = lr, mo, wd

    # ------------------------------------------------------------------ pieces
    def _drop(self, why, apply_pending=True):
        if apply_pending:
            self.flush()
        else:
            self.pending = None
        self.U.clear()
        self.V.clear()
        self.B.clear()
        self.warm.clear()

    def _abort_capture(self):
        try:
            torch.cuda.synchronize()
        except Exception:
            pass
        setattr(self.model.detector, '_features_override', None)
        self.model.__dict__['_seed_dev'] = None

    def _point_grads(self, b):
        for p, g in b.pgrads.items():
            p.grad = g

    def _static_like(self, batch):
        dev = batch[3].device
        st = list(batch)
        st[0] = [torch.empty_like(im) for im in batch[0]]
        for i in (3, 4, 5):
            st[i] = torch.empty_like(batch[i])
        return st

    def _feed(self, b, batch):
        """the batch into B's static inputs (one multi-tensor copy), the step's dropout seed into device memory"""
        srcs = list(batch[0]) + [batch[3], batch[4], batch[5]]
        dsts = list(b.static[0]) + [b.static[3], b.static[4], b.static[5]]
        if os.environ.get('SGG_GRAPH_DEBUG') == '2':
            import sys
            rng = lambda t: '%x-%x' % (t.data_ptr(), t.data_ptr() + t.numel() * t.element_size())
            sys.stderr.write('[graph] feed srcs %s dsts %s fmap %s seed %x\n' % (' '.join(rng(t) for t in srcs), ' '.join(rng(t) for t in dsts), rng(b.fmap) if hasattr(b, 'fmap') else '-', self.seed.data_ptr()))
            sys.stderr.flush()
        torch._foreach_copy_(dsts, srcs)
        _dbg('feed')
        for i in (4, 5):
            b.static[i]._sgg_host = getattr(batch[i], '_sgg_host', None)
        # the launch-by-launch step draws its seed from torch's CPU generator (train.predict_train): the same draw, so that n replayed steps
        # and n plain steps consume the same stream of seeds
        self.seed.fill_(int(torch.randint(0, 2 ** 31 - 1, (1,)).item()))

    def _wait_operands(self):
        ev = getattr(self.model, '_operands_ready', None)
        if ev is not None:
            torch.cuda.current_stream(self.seed.device).wait_event(ev)
            self.model._operands_ready = None

    LANE_SWITCHES = ('SGG_IMP_STREAMS', 'SGG_TRAIN_IMP_LANE', 'SGG_BWD_LANE')

    def _capture(self, fn, pool='main'):
        """fn's launches as ONE-STREAM graph: measured on this runtime (tools/graph_probe.py), a graph whose nodes span two streams costs
        its launch 2.6 ms of the calling thread (15 us per node, worse than launching the kernels one by one) and back-to-back launches of
        such graphs stall each other (12.5 ms per step in the bench's rotation); a one-stream graph of the same kernels launches in 0.07 ms.
        The lanes the launch-by-launch path forks inside the head (scheduling only, never a different sum) are therefore switched off for
        the capture; the overlap that matters -- the optimiser's update beside the VGG forward -- is two graphs on two streams."""
        dev = self.seed.device
        g = torch.cuda.CUDAGraph()
        with _lib.CAPTURE_LOCK, self._lanes_off():
            torch.cuda.synchronize(dev)
            with torch.cuda.graph(g, pool=self._pool(pool), capture_error_mode='relaxed'):
                out = fn()
        self.stats['captures'] += 1
        return g, out

    def _pool(self, which):
        """two memory pools: graphs that run on the calling stream ('main') and graphs that run on the node lane's stream beside them
        ('lane': the update, the backward's lane work) never recycle each other's temporaries"""
        if self.pool is None:
            self.pool = {'main': torch.cuda.graph_pool_handle(), 'lane': torch.cuda.graph_pool_handle()}
        return self.pool[which]

    def _lanes_off(self):
        import contextlib

        @contextlib.contextmanager
        def cm():
            saved = {k: os.environ.get(k) for k in self.LANE_SWITCHES}
            for k in self.LANE_SWITCHES:
                os.environ[k] = '0'
            try:
                yield
            finally:
                for k, v in saved.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
        return cm()

    def _capture_segments(self, fn):
        """fn's launches as a SEQUENCE of one-stream graphs cut where fn calls model._graph_split.next(tag) (train.PredictFn.backward:
        'lane' = the backward's lane work, 'main' = phases B and C, 'joined' = what follows their meeting point)
        -> ([(tag, graph)], fn's result)"""
        m = self.model
        dev = self.seed.device
        segs = []
        outer = self

        class Split(object):
            def begin(self, tag):
                self.tag, self.g = tag, torch.cuda.CUDAGraph()
                self.g.capture_begin(pool=outer._pool('lane' if tag == 'lane' else 'main'), capture_error_mode='relaxed')

            def end(self):
                with warnings.catch_warnings(record=True) as seen:
                    warnings.simplefilter('always')
                    self.g.capture_end()
                # 'joined' holds launches only when a caller wants input gradients.  An EMPTY graph is never launched: on this runtime (ROCm 7.0
                # HIP) replaying one ended in "Memory access fault by GPU" within a few dozen steps -- 5 of 5 short bench runs, 0 of 10 without
                # it (tools/graph_fault_stats.sh, round 5)
                empty = any('Graph is empty' in str(w_.message) for w_ in seen)
                segs.append((self.tag, None if (empty and os.environ.get('SGG_GRAPH_KEEP_EMPTY') != '1') else self.g))

            def next(self, tag):
                self.end()
                self.begin(tag)
        sp = Split()
        cap = torch.cuda.Stream(device=dev)
        with _lib.CAPTURE_LOCK, self._lanes_off():
            torch.cuda.synchronize(dev)
            cap.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(cap):
                sp.begin('head')
                m.__dict__['_graph_split'] = sp
                try:
                    out = fn()
                finally:
                    m.__dict__['_graph_split'] = None
                    sp.end()
            torch.cuda.current_stream(dev).wait_stream(cap)
        self.stats['captures'] += len(segs)
        return segs, out

    def _capture_B(self, sig, batch):
        tr, m = self.tr, self.model
        dev = batch[3].device
        # this step's update of the previous (plain) step is already queued on the lane: the head must wait for it as a plain step would --
        # outside the capture (an event of a stream that is not being captured)
        self.flush()
        self._wait_operands()
        from .train import train_weights
        train_weights(m)                  # (a no-op unless something changed the weights outside a step: nothing of it may land in the capture)
        if hasattr(m, 'check_pair_flag'):
            m.check_pair_flag(wait=True)
        b = _B()
        b.sig, b.replays = sig, 0
        b.static = self._static_like(batch)
        self._feed(b, batch)
        st = tuple(b.static)
        # the feature map at one address: this step's VGG forward launch by launch into it (later steps: graph A)
        det = m.detector
        probe, sizes, padded = det.features(st[0], m.compute_dtype)
        b.fmap = torch.empty_like(probe)
        b.fmap.copy_(probe)
        del probe
        b.sizes, b.padded = sizes, padded
        m.__dict__['_seed_dev'] = self.seed
        m.__dict__['_transposes_in_forward'] = self.defer_transposes
        det._features_override = (b.fmap, sizes, padded)
        try:
            def body():
                loss = tr._forward_backward(st)
                return loss, {p: p.grad for p in tr.opt.params() if p.grad is not None}, dict(tr._local)
            b.segments, (b.loss, b.pgrads, b.local) = self._capture_segments(body)
            tags = [t for t, _ in b.segments]
            if tags[0] != 'head' or any(t not in ('lane', 'main', 'joined') for t in tags[1:]) or tags.count('lane') != tags.count('joined'):
                raise RuntimeError('unexpected capture segments %s' % tags)
        finally:
            det._features_override = None
            m.__dict__['_seed_dev'] = None
            m.__dict__['_transposes_in_forward'] = False
        b.pins = pin_caches(m)          # the index tensors the captured launches read stay alive with the graph
        self.B[sig] = b
        return b

    def _run_A(self, b):
        """update of the pending step's gradients (graph U, on the node lane's stream)  ||  VGG forward of this step's images into b.fmap
        (graph V, on the calling stream): replayed, or captured first.  The head (graph B) waits for both."""
        tr, m = self.tr, self.model
        from .imp import node_lane
        from .train import train_weights
        dev = self.seed.device
        main = torch.cuda.current_stream(dev)
        prev = self.pending
        done, rebuild = None, None
        vkey = (id(b), str(m.compute_dtype), self.v_split)
        v = self.V.get(vkey)
        if v is None:
            v = self.V[vkey] = self._capture_V(b)
        for part in v.parts[:-1]:           # SGG_GRAPH_VSPLIT: the first layers before the update is released (it waits for this stream below)
            part.replay()
        if prev is None:
            self._wait_operands()       # the step before was a plain one: its update is queued on the lane, the head waits for it as a plain step would
        else:
            ukey = (id(prev), self._opt_scalars(), str(m.compute_dtype))
            u = self.U.get(ukey)
            fresh = u is None
            if fresh:
                self._point_grads(prev)
                # (SGG_GRAPH_SPLIT_UPDATE=1: U1 = the squared norms + clip + SGD on the lane's stream beside the VGG forward, U2 = the rebuild of
                # the weight-derived operands on the calling stream AFTER the two have met -- measured slower, see __init__)
                if self.split_update:
                    u1, _ = self._capture(lambda: tr.opt.step(grad_scale=1.0 / tr.loss_scale, grads=dict(prev.local), shards=None), pool='lane')
                    # (the optimiser's Python-side bookkeeping of ONE update ran during the capture: the replay below is that update)
                    u2, _ = self._capture(lambda: train_weights(m, transposes=not self.defer_transposes), pool='main')
                else:
                    def body():
                        tr.opt.step(grad_scale=1.0 / tr.loss_scale, grads=dict(prev.local), shards=None)
                        train_weights(m, transposes=not self.defer_transposes)
                    steps_before = tr.opt.steps
                    try:
                        u1, _ = self._capture(body, pool='lane')
                    except Exception:
                        tr.opt.steps = steps_before         # nothing a failed capture recorded has run: the host-side count of updates neither
                        raise
                    u2 = None
                u = self.U[ukey] = (u1, u2)
            lane = node_lane(dev)
            side = lane[0] if lane is not None else main
            if side is not main:
                side.wait_stream(main)          # the gradients (the previous step's graph B) and whatever else this stream has queued
                with torch.cuda.stream(side):
                    u[0].replay()
                    self.pending = None         # issued: whatever fails from here on, these gradients are never applied a second time
                    done = torch.cuda.Event()
                    done.record(side)
            else:
                u[0].replay()
                self.pending = None
            _dbg('U (update) %s' % ('captured' if fresh else 'replayed'))
            if not fresh:
                self._after_update()
            rebuild = u[1]
        v.parts[-1].replay()
        _dbg('V (VGG forward)')
        if done is not None:
            main.wait_event(done)
        if rebuild is not None:
            rebuild.replay()
        self.pending = None

    def _capture_V(self, b):
        """the VGG forward of b's static images into b.fmap: one graph, or two cut after the first `v_split` convolutions"""
        m = self.model
        det = m.detector
        fn = lambda: det.features(b.static[0], m.compute_dtype, out=b.fmap)
        k = self.v_split
        if k <= 0:
            return _Seq([self._capture(fn)[0]])
        det._features_split = lambda convs_done: m.__dict__['_graph_split'].next('main') if convs_done == k else None
        try:
            segs, _ = self._capture_segments(fn)
        finally:
            det._features_split = None
        return _Seq([g for _, g in segs if g is not None])

    def _after_update(self):
        """what FusedSGD.step / Trainer._bump do on the host for one update, for an update that ran inside a replayed graph"""
        tr = self.tr
        o = tr.opt
        o.steps += 1
        live = [p for p in o.params() if p.grad is not None or p in tr._local]
        torch.autograd.graph.increment_version(live)
        if o.on_update is not None:
            o.on_update()

    def _replay_B(self, b):
        tr, m = self.tr, self.model
        self._launch_B(b)
        b.replays += 1
        self._point_grads(b)
        tr._local = dict(b.local)
        self.pending = b
        self.pending_scalars = self._opt_scalars()
        tr._queued = False
        m.global_batch_iter = getattr(m, 'global_batch_iter', 0) + 1
        if b.replays > 1 and hasattr(m.union_boxes, 'count_train_batch'):
            m.union_boxes.count_train_batch()       # (the capture itself counted the first one)

    def _launch_B(self, b):
        """the segments in capture order: 'head' / 'joined' on the calling stream ('joined' after the lane's graph has finished), 'lane' on the
        node lane's stream beside the 'main' segment that follows it"""
        from .imp import node_lane
        dev = self.seed.device
        main = torch.cuda.current_stream(dev)
        lane = node_lane(dev)
        side = lane[0] if lane is not None else main
        done = None
        for tag, g in b.segments:
            if tag == 'lane' and side is not main:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    if g is not None:
                        g.replay()
                    done = torch.cuda.Event()
                    done.record(side)
                continue
            if tag == 'joined' and done is not None:
                main.wait_event(done)
                done = None
            if g is not None:
                g.replay()
        if done is not None:
            main.wait_event(done)
