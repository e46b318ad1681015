"""Train step of main.py:100-120 on the HIP path, data-parallel over one process per GPU.

    res  = model(batch)                                   # train-mode forward (sgg_amd/train.py)
    loss = node CE + edge CE, normalised by the GLOBAL batch (lib/losses.py:34-63,74)
    loss.backward()                                       # HIP backward of the head
    all-reduce(SUM) of the 247.75 M gradients over RCCL   # sgg_amd/dist.py
    global-norm clip (5.0) + SGD(momentum 0.9, wd 1e-4, LR/10 for roi_fmap*)   # fused HIP kernels, no host sync
"""
import os

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import _lib, ops
from .dist import GradBuckets, sum_over_ranks


class FusedSGD(torch.optim.Optimizer):
    """torch.optim.SGD(momentum, weight_decay) + clip_grad_norm (lib/pytorch_misc.py:625-656) as HIP kernels.

    A `torch.optim.Optimizer` with the parameter groups of get_optim (lib/pytorch_misc.py:128-144: group 0 = names starting with
    'roi_fmap' at lr/10, group 1 = the rest) and torch's per-parameter state ('momentum_buffer'), so that
    `optim.lr_scheduler.MultiStepLR(optimizer, ...)` (:152-154), `get_smallest_lr`, `save_checkpoint` / `optimizer.load_state_dict(ckpt
    ['optimizer'])` (:146-150, :218-227) work on it unchanged and optimiser checkpoints interchange with the reference's."""

    def __init__(self, named_params, lr, momentum=0.9, weight_decay=1e-4, clip=5.0):
        named = [(n, p) for n, p in named_params if p.requires_grad]
        fc = [p for n, p in named if n.startswith('roi_fmap')]
        rest = [p for n, p in named if not n.startswith('roi_fmap')]
        groups = ([{'params': fc, 'lr': lr / 10.0}] if fc else []) + ([{'params': rest}] if rest else [])
        super(FusedSGD, self).__init__(groups, dict(lr=lr, momentum=momentum, dampening=0, weight_decay=weight_decay, nesterov=False))
        self.name_of = {p: n for n, p in named}
        self.clip = clip
        self.steps = 0
        self._norm = None
        self._norm_parts = None
        self._norm_ws = None
        self.sync_norm = False     # data-parallel: every rank clips with the SAME squared norm (Trainer turns it on)
        self.stale_masters = {}    # sharded data-parallel steps: {param: (lo, hi)} whose fp32 master is current only in [lo, hi)
        self.momentum_parts = {}   # ... and whose momentum buffer is
        self.on_update = None
        self.max_blocks = 0        # workgroups of the update kernel (0: library default); the pipelined trainer lowers it
        self.shadow_of = None      # callable -> {param name: compute-dtype buffer the update should also write}
        self.skipped = None        # i32[1] on the device: steps the kernels skipped because the gradient norm was not finite
        self.wrote_shadow = []

    def params(self):
        for g in self.param_groups:
            for p in g['params']:
                yield p

    @torch.no_grad()
    def step(self, grad_scale=1.0, grads=None, closure=None, shards=None):
        """grads (optional): {param: tensor} overriding p.grad -- e.g. the bf16 buffers an all-reduce left behind, consumed
        directly (fp32 or bf16) instead of being copied back into fp32 .grad tensors.
        shards (optional, data-parallel): {param: (lo, hi)} -- grads[param] is the reduce-scattered gradient of elements
        [lo, hi) of the flattened parameter only (GradBuckets(shard=True)).  This rank updates that part of the fp32 master and
        of the momentum buffer; the squared norms of the parts are summed over the ranks for the clip; the updated operand (the
        bf16 shadow when there is one, else the fp32 parameter itself) is all-gathered in place.  The fp32 master outside
        [lo, hi) is then STALE on this rank until gather_masters()."""
        import numpy as np
        stream = ops._stream()
        dev = next(self.params()).device
        if self._norm is None:
            self._norm = torch.zeros(1, dtype=torch.float32, device=dev)
            self._norm_ws = torch.empty(2048, dtype=torch.float32, device=dev)      # partial sums of sgg_sqnorm_multi (fixed-order reduction)
            self.skipped = torch.zeros(1, dtype=torch.int32, device=dev)
        norm_started = False                                    # the first sgg_sqnorm_multi of the step overwrites the accumulator
        grads = grads or {}
        shards = {p: r for p, r in (shards or {}).items() if r is not None}
        live = []                                               # (param, gradient, lr)
        mom, wd = None, None
        for g in self.param_groups:
            if g['dampening'] != 0 or g['nesterov']:
                raise NotImplementedError('FusedSGD: dampening / nesterov')
            mom = g['momentum'] if mom is None else mom
            wd = g['weight_decay'] if wd is None else wd
            if g['momentum'] != mom or g['weight_decay'] != wd:
                raise NotImplementedError('FusedSGD: one momentum / weight decay for all groups (as get_optim builds them)')
            for p in g['params']:
                gr = grads.get(p, p.grad)
                if gr is not None:
                    live.append((p, gr.contiguous(), float(g['lr'])))
        if not live:
            return None
        # torch's first step sets buf = d_p; a zero buffer gives the same number (mom * 0 + d_p), so parameters that meet their
        # first gradient later -- or after a partial load_state_dict -- simply start from zeros
        # (always zeros, never `empty` + a "first step" kernel flag: a first step that the kernels SKIP -- f16 overflow at the loss scale, a
        # NaN batch -- would otherwise leave the buffer uninitialised for the second step's mom * buf + g)
        first = False
        for p, _, _ in live:
            st = self.state[p]
            if st.get('momentum_buffer') is None:
                st['momentum_buffer'] = torch.zeros_like(p, dtype=torch.float32)
        shadows = self.shadow_of() if self.shadow_of is not None else {}
        norm = self._norm.data_ptr() if self.clip and self.clip > 0 else None
        for p, gr, _ in live:
            if p in shards and gr.numel() != shards[p][1] - shards[p][0]:
                raise ValueError('FusedSGD: the gradient of a sharded parameter must be its [lo, hi) part')
        if shards and norm is not None and self._norm_parts is None:
            self._norm_parts = torch.zeros(1, dtype=torch.float32, device=dev)
        pending = []
        for part in (False, True):                              # whole tensors first, then this rank's parts of the sharded ones
            for dtype in (torch.float32, torch.bfloat16, torch.float16):      # gradients arrive fp32 (local) or 16-bit (off the wire)
                sel = [t for t in live if t[1].dtype == dtype and (t[0] in shards) == part]
                if not sel:
                    continue
                arr = lambda vals: np.ascontiguousarray(np.array(vals, dtype=np.uint64))
                lo = [shards[p][0] if part else 0 for p, _, _ in sel]
                gp = arr([gr.data_ptr() for _, gr, _ in sel])
                pp = arr([p.data_ptr() + 4 * o for (p, _, _), o in zip(sel, lo)])
                bp = arr([self.state[p]['momentum_buffer'].data_ptr() + 4 * o for (p, _, _), o in zip(sel, lo)])
                sh = [shadows.get(self.name_of.get(p)) for p, _, _ in sel]
                sp = arr([t.data_ptr() + t.element_size() * o if t is not None else 0 for t, o in zip(sh, lo)])
                nn = np.ascontiguousarray(np.array([gr.numel() for _, gr, _ in sel], dtype=np.int64))
                lr = np.ascontiguousarray(np.array([l for _, _, l in sel], dtype=np.float32))
                if norm is not None:
                    acc = self._norm_parts.data_ptr() if part else norm
                    _lib.call('sgg_sqnorm_multi', gp.ctypes.data, nn.ctypes.data, len(sel), acc, self._norm_ws.data_ptr(),
                              1 if (part or norm_started) else 0, ops.dt(sel[0][1]), stream)
                    norm_started = norm_started or not part
                pending.append((dtype, gp, pp, bp, sp, nn, lr, len(sel), sel))
        if norm is not None and not norm_started:
            self._norm.zero_()                                  # every live tensor is sharded: the replicated share is 0
        if norm is not None and (shards or self.sync_norm):
            # the parts' squared norms of all ranks join the replicated tensors' one, and EVERY rank ends up with the same number: the
            # replicated share travels as norm / world (the reduced gradients are the same bits on every rank and the local reduction
            # order is fixed, so the shares are equal and their sum is the local value again -- exactly, for power-of-two worlds; ranks
            # that disagreed in a last bit would get the mean instead of drifting apart), the parts as they are
            import torch.distributed as dist
            world = dist.get_world_size()
            pair = torch.stack((self._norm[0] / world, self._norm_parts[0] if self._norm_parts is not None else self._norm.new_zeros(())))
            dist.all_reduce(pair, op=dist.ReduceOp.SUM)
            self._norm.copy_((pair[0] + pair[1]).view(1))
            if self._norm_parts is not None:
                self._norm_parts.zero_()
        for k, (dtype, gp, pp, bp, sp, nn, lr, cnt, _keep) in enumerate(pending):  # every norm contribution lands before the first update
            _lib.call('sgg_sgd_multi', pp.ctypes.data, gp.ctypes.data, bp.ctypes.data, sp.ctypes.data, nn.ctypes.data,
                      lr.ctypes.data, cnt, float(wd), float(mom), int(first), norm,
                      float(self.clip or 0.0), float(grad_scale), ops.dt(dtype), self._shadow_dt(shadows), int(self.max_blocks),
                      self.skipped.data_ptr() if k == 0 else None, stream)
        self.gather_bytes_last = 0
        if shards:                                              # every rank gets every part of the updated operands
            import torch.distributed as dist
            for p, _, _ in live:
                if p in shards:
                    sh = shadows.get(self.name_of.get(p))
                    full = (sh if sh is not None else p.data).view(-1)
                    dist.all_gather_into_tensor(full, full[shards[p][0]:shards[p][1]])
                    self.gather_bytes_last += full.numel() * full.element_size()
            self.stale_masters = {p: shards[p] for p, _, _ in live if p in shards and shadows.get(self.name_of.get(p)) is not None}
            self.momentum_parts = {p: shards[p] for p, _, _ in live if p in shards}
        names = set(self.name_of.get(p) for p, _, _ in live)
        self.wrote_shadow = [n for n in shadows if n in names]
        self.steps += 1
        # the kernels wrote through raw pointers: move every updated parameter's autograd version counter, so that anything keyed
        # on (data_ptr, _version) -- model.prepared(), the bf16 shadows, the W^T copies of the backward -- sees the change even
        # when this optimiser is used on its own (scheduler / checkpoint interchange) without a Trainer's on_update
        torch.autograd.graph.increment_version([p for p, _, _ in live])
        if self.on_update is not None:
            self.on_update()   # parameters changed through raw pointers: tell the owner to refresh derived operands
        return None

    @staticmethod
    def _shadow_dt(shadows):
        for t in shadows.values():
            if t is not None:
                return ops.dt(t)
        return _lib.SGG_BF16

    def grad_norm(self, grad_scale=1.0):
        """global gradient norm of the last step (of the UNSCALED gradients when the step's grad_scale is passed)"""
        return float(self._norm.sqrt().item()) * float(grad_scale)

    @torch.no_grad()
    def gather_masters(self, momentum=True):
        """After sharded steps: all-gather the fp32 masters (and momentum buffers) whose other parts went stale on this rank,
        so that checkpoints, fp32-mode forwards and `state_dict()` see whole tensors again.  A collective: every rank calls it."""
        import torch.distributed as dist
        for p, (lo, hi) in self.stale_masters.items():
            flat = p.data.view(-1)
            dist.all_gather_into_tensor(flat, flat[lo:hi])
        if momentum:
            for p, (lo, hi) in self.momentum_parts.items():
                buf = self.state[p].get('momentum_buffer')
                if buf is not None:
                    flat = buf.view(-1)
                    dist.all_gather_into_tensor(flat, flat[lo:hi])
            self.momentum_parts = {}
        if self.stale_masters:
            torch.autograd.graph.increment_version(list(self.stale_masters))
        self.stale_masters = {}


class Trainer(object):
    """One data-parallel train step.  `loss_type` as lib/losses.py ('baseline' is the reference default, config.py:184)."""

    def __init__(self, model, lr=1e-3, momentum=0.9, weight_decay=1e-4, clip=5.0, loss_type='baseline',
                 comm_dtype=torch.bfloat16, force_dist=False, sync_bn=True, pipeline=False, loss_weights=(1, 1, 1),
                 shard_optimizer=None, graph=None):
        self.model = model
        for n, p in model.named_parameters():
            if n.startswith('detector.'):
                p.requires_grad = False                     # main.py:62-63: the detector is frozen
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        self.opt = FusedSGD(named, lr, momentum, weight_decay, clip)
        self.opt.on_update = self._bump
        if hasattr(model, 'shadow_buffers'):
            self.opt.shadow_of = model.shadow_buffers
        # shard_optimizer (default: on when there is more than one rank; SGG_SHARD_OPT=0 turns it off): the four tensors that are
        # 97 % of the parameters (fc6 x2, fc7 x2) are reduce-scattered, each rank runs clip + SGD on its 1/world of their fp32
        # masters and momenta, and the updated compute-dtype operands are all-gathered -- the bytes of the all-reduce on the
        # links, 1/world of the 5 GB optimiser pass per rank.  fp32 masters are whole again after flush().
        n_ranks = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if shard_optimizer is None:
            shard_optimizer = n_ranks > 1 and os.environ.get('SGG_SHARD_OPT', '1') != '0'
        self.shard_optimizer = bool(shard_optimizer) and (n_ranks > 1 or (force_dist and dist.is_available() and dist.is_initialized()))
        self.buckets = GradBuckets([p for _, p in named], comm_dtype=comm_dtype, force=force_dist, shard=self.shard_optimizer)
        if self.shard_optimizer:
            model._sharded_group_sum = self._group_sum_of_parts
        self.loss_type = loss_type
        self.loss_weights = tuple(float(x) for x in loss_weights)      # (alpha, beta, gamma) of lib/losses.py:37 = conf.alpha/beta/gamma
        # pipeline=True: the optimiser update of step k (HBM-bound, ~1.1 ms) and the rebuild of the weight-derived operands
        # run on the side stream UNDER the frozen VGG forward of step k+1 (MFMA-bound, ~2 ms), which does not read the
        # weights being updated (main.py:62-63: the detector is frozen).  The head of step k+1 waits for the event.
        # Every update still happens, in order; read parameters through flush() in this mode.
        self.pipeline = pipeline
        self.fused_loss = True          # step(): loss + logit gradients by sgg_ce_fwd_bwd where it applies (tests flip it to compare)
        # f16 compute: activation gradients are 16-bit tensors with five exponent bits -- the loss is scaled by `loss_scale` on the way
        # into the backward (every backward kernel is linear in the incoming gradient) and the optimiser divides it out again
        # (grad_scale of sgg_sgd_multi, applied in fp32; the clip works on the unscaled norm).  A step whose scaled gradients overflow
        # has a non-finite norm: the update kernels skip it.  bf16 / f32: scale 1.
        self.loss_scale_f16 = float(os.environ.get('SGG_LOSS_SCALE', '1024'))
        self._norm_cache = {}
        if pipeline:
            # leave wave slots for the VGG forward running beside the update (round 5, after the update kernel's loads were un-serialised:
            # 192 workgroups 6.81 - 6.88 ms per step, 256: 6.87 - 6.94, same box)
            self.opt.max_blocks = int(os.environ.get('SGG_OPT_BLOCKS', '192'))
        self._queued = False
        self.graphs = None        # sgg_amd.graph_step.GraphStep when the step is replayed as hipGraphs (set at the end of __init__)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # force_dist: run the distributed code path on a 1-rank group too (tests exercise RCCL plumbing on one GPU)
        self.dist_on = self.world > 1 or (force_dist and dist.is_available() and dist.is_initialized())
        self._local = {}          # one GPU: wire-dtype gradients of the big tensors, kept out of autograd for the optimiser
        self._local_wire = comm_dtype if comm_dtype == torch.bfloat16 else None
        self.opt.sync_norm = self.dist_on and self.world > 1
        if not self.dist_on and self._local_wire is not None:
            by_name_l = dict(named)

            def keep(name, grad):     # same hand-over as the data-parallel hook, without a collective: with bf16 compute the
                p = by_name_l.get(name)   # weight-gradient GEMMs of fc6 / fc7 emit bf16 (fp32 accumulate, one rounding) and the
                if p is not None and self.buckets.is_big(p):   # optimiser reads that -- 1-GPU and N-GPU steps see the same numbers,
                    self._local[p] = grad                      # and 0.8 GB less is written and re-read per step
                    return True
                return False
            self._keep = keep         # installed on the model only for the duration of step()'s backward
        if self.dist_on:
            by_name = dict(named)

            def early(name, grad):      # called from PredictFn.backward the moment a big gradient exists
                p = by_name.get(name)
                if p is not None and self.buckets.is_big(p):
                    self.buckets.start(p, grad)
                    return True
                return False
            model._grad_ready_hook = early
            # bf16 on the wire: the weight-gradient GEMMs of the hooked tensors emit bf16 directly (fp32 accumulate,
            # one rounding -- the same numbers as casting an fp32 gradient, without writing and re-reading it)
            model._grad_wire_dtype = comm_dtype if comm_dtype == torch.bfloat16 else None
            # union_boxes.conv BatchNorm: statistics over the edges of the GLOBAL batch, i.e. the numbers a
            # single process would compute on the concatenated batch; sync_bn=False = replica-local statistics.
            model._bn_sync = (lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM)) if sync_bn else None
        # graph (default: on for the pipelined one-GPU trainer; SGG_GRAPH=0 turns it off): the step replayed as hipGraphs wherever the
        # batch allows it (sgg_amd/graph_step.py) -- same kernels, same arguments, same order; ~0.3 ms of host time per step instead of 3.5 - 5.5
        if graph is None:
            graph = os.environ.get('SGG_GRAPH', '1') != '0'
        if graph and pipeline and not self.dist_on and torch.cuda.is_available():
            from .graph_step import GraphStep
            self.graphs = GraphStep(self)

    def local_only(self):
        """Context manager: inside it the trainer and the model issue NO collective (gradient hooks, BatchNorm statistics,
        loss normalisers, bucket all-reduce) -- for extra steps that one rank runs on its own, e.g. a profiling pass while the
        other ranks wait.  Everything is restored on exit."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            m = self.model
            saved = (getattr(m, '_grad_ready_hook', None), getattr(m, '_bn_sync', None), self.world, self.dist_on)
            saved_sh = (getattr(m, '_sharded_group_sum', None), self.opt.stale_masters, self.opt.momentum_parts)
            m._grad_ready_hook = m._bn_sync = m._sharded_group_sum = None
            self.world, self.dist_on = 1, False
            # (with a sharded optimiser the fp32 masters this rank does not own are stale in here: fine for timing, not for results)
            self.opt.stale_masters, self.opt.momentum_parts = {}, {}
            try:
                yield self
            finally:
                m._grad_ready_hook, m._bn_sync, self.world, self.dist_on = saved
                m._sharded_group_sum, self.opt.stale_masters, self.opt.momentum_parts = saved_sh
        return cm()

    def _group_sum_of_parts(self, w6e, C, PP, dtype):
        """W6sum (model.prepared()) when this rank's fp32 master of fc6 is current only in its own rows: every rank sums its rows,
        the [rows, C] results are all-gathered -- the numbers of the whole-tensor sum.  A collective: prepared() runs at the same
        points of the step on every rank."""
        p = self.model.fc_layers()[0]['fc6_edge'][1].weight
        rng = self.opt.stale_masters.get(p) if p is not None else None
        if rng is None:
            return ops.group_sum(w6e, C, PP, dtype)
        rows, K = w6e.shape
        lo, hi = rng
        if lo % K or hi % K:
            raise ValueError('sharded optimiser: the parts of fc6 must be whole rows (world size must divide %d)' % rows)
        part = ops.group_sum(w6e[lo // K:hi // K], C, PP, dtype)
        full = torch.empty((rows, part.shape[1]), dtype=part.dtype, device=part.device)
        dist.all_gather_into_tensor(full.view(-1), part.reshape(-1))
        return full

    @property
    def loss_scale(self):
        # f16 compute, and the x3 mode: its contractions split every operand -- the backward's dY included -- into f16 halves (five exponent
        # bits: |x| < 6e-8 has no hi half, |x| < 0.125 a subnormal lo half), and head gradients at the bench size are 1e-6 .. 1e-8 per
        # element.  Same mechanism as f16: scaled loss in, grad_scale out, a step whose scaled gradients overflow is skipped.
        if self.model.compute_dtype == torch.float16 or getattr(self.model, 'split3', False):
            return self.loss_scale_f16
        return 1.0

    def _shards(self, reduced):
        return {p: self.buckets.shard_of(p) for p in reduced} if self.shard_optimizer else None

    def _bump(self):
        self.model.weights_version = getattr(self.model, 'weights_version', 0) + 1
        if hasattr(self.model, 'mark_shadow_fresh'):
            self.model.mark_shadow_fresh(self.opt.wrote_shadow)

    def losses(self, res):
        """node + edge classification losses with GLOBAL-batch normalisers: summed local CE / global counts, so that
        the all-reduce(SUM) of the gradients equals the single-process gradient on the concatenated batch."""
        obj_ce = F.cross_entropy(res.rm_obj_dists, res.rm_obj_labels, reduction='sum')
        labels = res.rel_labels[:, -1]
        rel_ce = F.cross_entropy(res.rel_dists, labels, reduction='none')
        n_obj, M = float(res.rm_obj_labels.shape[0]), float(labels.shape[0])
        alpha, beta, gamma = self.loss_weights
        if self.loss_type == 'baseline':
            assert alpha == beta == 1, ('wrong loss is used, use dnorm or dnorm-fgbg', alpha, beta)      # lib/losses.py:41
            if self.dist_on:
                # global normalisers stay on the device (fill_ = a launch, no host round trip): a .tolist() here would
                # stall the host between forward and backward on every rank
                t = torch.empty(2, dtype=torch.float32, device=res.rel_dists.device)
                t[0].fill_(n_obj)
                t[1].fill_(M)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                return obj_ce / t[0] + gamma * rel_ce.sum() / t[1]
            return obj_ce / n_obj + gamma * rel_ce.sum() / M                         # lib/losses.py:41-43,74
        if self.loss_type not in ('dnorm', 'dnorm-fgbg'):
            raise NotImplementedError(self.loss_type)
        # density-normalised forms (lib/losses.py:44-63): M_FG / M_BG stay ON THE DEVICE -- counted, summed over the ranks and turned into
        # the row weights there (the reference's len(idx_fg) is a host number; a .item() here would stall the host between forward and
        # backward on every step).  The reference's "stay 1" branches (M_FG == 0, M_BG == 0) are torch.where selections.
        fg = labels > 0
        cnt = torch.stack((fg.sum(), (labels == 0).sum())).to(torch.float32)
        if self.dist_on:
            t = torch.cat((cnt.new_full((1,), n_obj), cnt))
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            n_obj, cnt = t[0], t[1:]
        m_fg, m_bg = cnt[0], cnt[1]
        one = torch.ones((), dtype=rel_ce.dtype, device=rel_ce.device)
        w_fg = torch.where(m_fg > 0, alpha / m_fg.clamp(min=1), one)                  # :50-51
        if self.loss_type == 'dnorm':
            w_bg = torch.where((m_bg > 0) & (m_fg > 0), beta / m_fg.clamp(min=1), one)    # :56-57
        else:
            w_bg = torch.where(m_bg > 0, beta / m_bg.clamp(min=1), one)               # :59-60
        return obj_ce / n_obj + (gamma * rel_ce * torch.where(fg, w_fg, w_bg)).sum()  # :62-63

    def _fused_losses(self, res):
        """All three loss forms of lib/losses.py and their logit gradients in two launches (sgg_ce_fwd_bwd; + one counting launch for the
        density-normalised forms) instead of torch's ~25: returns the loss (device scalar, no autograd graph) after leaving the zero-padded
        compute-dtype gradients where PredictFn.backward picks them up (model._logit_grads).  Same numbers as losses() + autograd: sum CE /
        global normaliser.  Nothing here reads a device value on the host: M_FG / M_BG of 'dnorm' / 'dnorm-fgbg' are counted by
        sgg_label_counts, all-reduced as a device tensor and applied per row inside the kernel."""
        m = self.model
        dev = res.rel_dists.device
        N, E = res.rm_obj_dists.shape[0], res.rel_dists.shape[0]
        alpha, beta, gamma = self.loss_weights
        dnorm = self.loss_type != 'baseline'
        if not dnorm:
            assert alpha == beta == 1, ('wrong loss is used, use dnorm or dnorm-fgbg', alpha, beta)      # lib/losses.py:41
        labels = res.rel_labels[:, -1]
        if self.dist_on:
            norm = torch.empty(3, dtype=torch.float32, device=dev)       # [N | E] or [N | M_FG | M_BG], summed over the ranks
            norm[0].fill_(float(N))
            if dnorm:
                ops.label_counts(labels, norm[1:3])
            else:
                norm[1].fill_(float(E))
            dist.all_reduce(norm, op=dist.ReduceOp.SUM)
        elif dnorm:
            norm = torch.empty(3, dtype=torch.float32, device=dev)
            norm[0].fill_(float(N))
            ops.label_counts(labels, norm[1:3])
        else:
            norm = self._norm_cache.get((N, E))
            if norm is None:
                norm = self._norm_cache[(N, E)] = torch.tensor([float(N), float(E), 0.0], dtype=torch.float32, device=dev)
        dt_ = m.compute_dtype
        loss = torch.empty(1, dtype=torch.float32, device=dev)          # written by the first head's call, added to by the second
        d_obj = torch.empty((N, 256), dtype=dt_, device=dev)           # 151 -> 256, 51 -> 128: what the TN weight-gradient kernel takes
        d_rel = torch.empty((E, 128), dtype=dt_, device=dev)
        if getattr(self, '_label_flag', None) is None:
            self._label_flag = torch.zeros(1, dtype=torch.int32, device=dev)
        ops.ce_fwd_bwd(res.rm_obj_dists.detach(), res.rm_obj_labels, norm[0:1], 1.0, loss, d_obj, self.loss_scale, self._label_flag, accumulate=False)
        ops.ce_fwd_bwd(res.rel_dists.detach(), labels, norm[1:3], gamma, loss, d_rel, self.loss_scale, self._label_flag,
                       mode=self.loss_type, alpha=alpha, beta=beta)
        m._logit_grads = (d_obj, d_rel)
        return loss[0]

    def _prefetch_operands(self):
        """The optimiser just changed the masters, so the derived operands (W^T copies for the dX GEMMs, W6sum, the GRU /
        rect-conv operands: ~0.3 ms of HBM-bound transposes and casts) must be rebuilt.  They are rebuilt on the node
        lane's stream while the main stream runs the (MFMA-bound, frozen) VGG forward; PredictFn.forward waits for the
        event.  The side stream first waits for everything already queued on the main stream: the previous step may
        still be reading the buffers this rebuild recycles."""
        from .imp import node_lane
        from .train import train_weights
        dev = next(self.model.parameters()).device
        lane = node_lane(dev)
        if lane is None:
            return
        side = lane[0]
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            train_weights(self.model)
            ev = torch.cuda.Event()
            ev.record(side)
        self.model._operands_ready = ev

    def flush(self):
        """Make the current stream wait for a queued optimiser update (pipeline mode) before parameters are read; with a sharded
        optimiser also gather the fp32 masters and momentum buffers (a collective: every rank calls flush())."""
        if self.graphs is not None:
            self.graphs.flush()          # the last replayed step's update has not been applied yet (it rides in the NEXT step's first graph)
        ev = getattr(self.model, '_operands_ready', None)
        if ev is not None:
            torch.cuda.current_stream(next(self.model.parameters()).device).wait_event(ev)
        if self.dist_on and (self.opt.stale_masters or self.opt.momentum_parts):
            self.opt.gather_masters()
        if hasattr(getattr(self.model, 'union_boxes', None), 'flush_batch_counts'):
            self.model.union_boxes.flush_batch_counts()
        if hasattr(self.model, 'check_pair_flag'):
            self.model.check_pair_flag(wait=True)     # the last steps' pair-table flags (rel_model_stanford._watch_pair_flag)
        if self.opt.skipped is not None and self.loss_scale != 1.0:
            # f16: steps whose scaled gradients overflowed were skipped by the update kernels (counted on the device).  A static scale that
            # overflows on EVERY step would otherwise skip silently while `steps` and the LR schedule advance: halve it and say so.
            n_skipped, n_steps = int(self.opt.skipped.item()), self.opt.steps - getattr(self, '_steps_at_flush', 0)
            self.skipped_steps = getattr(self, 'skipped_steps', 0) + n_skipped
            self._steps_at_flush = self.opt.steps
            self.opt.skipped.zero_()
            if n_skipped:
                import warnings
                msg = 'sgg_amd: %d of the last %d train steps were skipped (non-finite gradients at loss scale %g)' % (n_skipped, n_steps, self.loss_scale_f16)
                if n_steps > 0 and 2 * n_skipped >= n_steps and self.loss_scale_f16 > 1.0:
                    self.loss_scale_f16 = max(1.0, self.loss_scale_f16 / 2.0)
                    msg += '; loss scale lowered to %g' % self.loss_scale_f16
                warnings.warn(msg, RuntimeWarning)
        flag = getattr(self, '_label_flag', None)
        if flag is not None and int(flag.item()) != 0:
            flag.zero_()
            raise ValueError('sgg_amd: a class / predicate label outside [0, C) reached the fused cross-entropy since the last flush() '
                             '(such rows added no loss and got zero gradients; F.cross_entropy\'s ignore_index behaves like that, any other '
                             'value is a data error)')

    def _queue_update(self):
        """pipeline mode: wait for the gradient all-reduce, optimiser step and operand rebuild, all on the side stream
        and after everything this step queued -- the main stream goes straight on to the next step's VGG forward, so
        the tail of the all-reduce hides under it as well."""
        from .imp import node_lane
        from .train import train_weights
        dev = next(self.model.parameters()).device
        lane = node_lane(dev)
        if lane is None:
            return False
        side = lane[0]
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for p in self.opt.params():                    # gradients were allocated on the main stream: keep their memory
                if p.grad is not None:                     # from being recycled there while the side stream reads it
                    p.grad.record_stream(side)
            reduced = self.buckets.all_reduce(average=False) if self.dist_on else dict(self._local)
            for t in (reduced or {}).values():
                t.record_stream(side)
            self.opt.step(grad_scale=1.0 / self.loss_scale, grads=reduced, shards=self._shards(reduced) if self.dist_on else None)
            train_weights(self.model)
            ev = torch.cuda.Event()
            ev.record(side)
        self.model._operands_ready = ev
        return True

    def __del__(self):
        # the objects frozen by step() (gc.freeze) go back under the collector's eyes with their trainer: a process that builds many
        # models (test suites, sweeps) must not keep their reference cycles alive for good
        if getattr(self, '_gc_frozen', False):
            try:
                import gc
                gc.unfreeze()
            except Exception:
                pass

    def update(self):
        """The tail of a step for gradients that already exist (p.grad, the early-started buckets of the backward's hooks, the kept
        wire-dtype gradients of the one-GPU path): reduce over the ranks, global-norm clip, SGD -- on the current stream.  step() ends
        with it; callers that run their own backward through the model (the GAN iteration's reconstruction losses, feature_gan.
        gan_train_step) call it instead of a plain optimiser's step()."""
        reduced = self.buckets.all_reduce(average=False) if self.dist_on else dict(self._local)
        self.opt.step(grad_scale=1.0 / self.loss_scale, grads=reduced, shards=self._shards(reduced) if self.dist_on else None)
        self._local = {}

    def _forward_backward(self, batch):
        """forward, loss and backward of one batch on the current stream -> the loss (device scalar); the gradients are left in p.grad
        and, on one GPU with a 16-bit wire type, in self._local (the big tensors, kept out of autograd for the optimiser)."""
        local = not self.dist_on and getattr(self, '_keep', None) is not None and ops.is_half(self.model.compute_dtype)
        if not self.dist_on:
            self._local = {}
        res = self.model([batch])
        # the fused loss covers the three loss forms of lib/losses.py in the 16-bit modes, logits straight out of predict()
        fused = (self.fused_loss and self.loss_type in ops.CE_MODES and ops.is_half(self.model.compute_dtype) and
                 not getattr(self.model, 'use_bias', False) and res.rm_obj_dists.shape[1] <= 256 and res.rel_dists.shape[1] <= 128)
        loss = self._fused_losses(res) if fused else self.losses(res)
        for p_ in self.opt.params():     # == zero_grad(set_to_none=True) without the optimiser's per-call bookkeeping
            p_.grad = None
        if local:
            self.model._grad_ready_hook, self.model._grad_wire_dtype = self._keep, self._local_wire
        self.model._loss_scaled = True          # this backward carries the f16 loss scale (train.py warns about one that does not)
        try:
            # the backward of the head is ONE Python function (train.PredictFn.backward): run it on this thread instead of handing it to
            # autograd's device thread and waiting for it (SGG_BWD_THREAD=1 restores the engine's default)
            with torch.autograd.set_multithreading_enabled(os.environ.get('SGG_BWD_THREAD', '0') == '1'):
                if fused:
                    # the real gradients wait in model._logit_grads; autograd only needs placeholders of the outputs' shape (no launch)
                    torch.autograd.backward([res.rm_obj_dists, res.rel_dists],
                                            [torch.empty_like(res.rm_obj_dists), torch.empty_like(res.rel_dists)])
                else:
                    (loss * self.loss_scale if self.loss_scale != 1.0 else loss).backward()
        finally:
            self.model._logit_grads = None
            self.model._loss_scaled = False
            if local:
                self.model._grad_ready_hook = self.model._grad_wire_dtype = None
        return loss

    def step(self, batch):
        if not self.model.training:      # (a walk over ~90 modules and their __setattr__: 0.35 ms of the issuing thread per step when repeated)
            self.model.train()
        if self.graphs is not None:
            # one GPU, 16-bit, pipelined: the step as two replayed hipGraphs per batch signature (sgg_amd/graph_step.py); None = this batch
            # runs through the launch-by-launch path below (warm-up of a new signature, shapes the capture does not take, ...)
            out = self.graphs.step(batch)
            if out is not None:
                return out
        if not self._queued:
            self._prefetch_operands()
        loss = self._forward_backward(batch)
        self._queued = self.pipeline and self._queue_update()
        if not self._queued:
            self.update()
        self.model.global_batch_iter = getattr(self.model, 'global_batch_iter', 0) + 1
        if self.opt.steps == 3 and not getattr(self, '_gc_frozen', False) and os.environ.get('SGG_GC_FREEZE', '1') != '0':
            # everything that lives as long as the model exists by now (modules, parameters, operand caches, index tables): out of the
            # cyclic collector's sight, so that a full collection -- measured as ONE 80-110 ms pause of the issuing thread per ~100 steps,
            # i.e. the GPU idle for most of it -- only walks what the steps themselves leave behind
            import gc
            gc.collect()
            gc.freeze()
            self._gc_frozen = True
        return loss.detach()
