"""Train step of main.py:100-120 on the HIP path, data-parallel over one process per GPU.

    res  = model(batch)                                   # train-mode forward (sgg_amd/train.py)
    loss = node CE + edge CE, normalised by the GLOBAL batch (lib/losses.py:34-63,74)
    loss.backward()                                       # HIP backward of the head
    all-reduce(SUM) of the 247.75 M gradients over RCCL   # sgg_amd/dist.py
    global-norm clip (5.0) + SGD(momentum 0.9, wd 1e-4, LR/10 for roi_fmap*)   # fused HIP kernels, no host sync
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import _lib, ops
from .dist import GradBuckets, sum_over_ranks


class FusedSGD(object):
    """torch.optim.SGD(momentum, weight_decay) + clip_grad_norm (lib/pytorch_misc.py:625-656) as HIP kernels.
    Same parameter groups as get_optim (lib/pytorch_misc.py:135-144): names starting with 'roi_fmap' get lr/10."""

    def __init__(self, named_params, lr, momentum=0.9, weight_decay=1e-4, clip=5.0):
        self.groups = []
        for n, p in named_params:
            if p.requires_grad:
                self.groups.append(dict(name=n, p=p, lr=lr / 10.0 if n.startswith('roi_fmap') else lr, buf=None))
        self.momentum, self.weight_decay, self.clip = momentum, weight_decay, clip
        self.steps = 0
        self._norm = None
        self.on_update = None
        self.max_blocks = 0        # workgroups of the update kernel (0: library default); the pipelined trainer lowers it
        self.shadow_of = None      # callable -> {param name: compute-dtype buffer the update should also write}
        self.wrote_shadow = []

    def zero_grad(self):
        for g in self.groups:
            g['p'].grad = None

    @torch.no_grad()
    def step(self, grad_scale=1.0, grads=None):
        """grads (optional): {param: tensor} overriding p.grad -- e.g. the bf16 buffers an all-reduce left behind, consumed
        directly (fp32 or bf16) instead of being copied back into fp32 .grad tensors."""
        stream = torch.cuda.current_stream().cuda_stream
        dev = self.groups[0]['p'].device
        if self._norm is None:
            self._norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._norm.zero_()
        grads = grads or {}
        grad_of = lambda p: grads.get(p, p.grad)
        live = [g for g in self.groups if grad_of(g['p']) is not None]
        import numpy as np
        first = any(g['buf'] is None for g in live)
        if first and not all(g['buf'] is None for g in live):
            raise RuntimeError('FusedSGD: a parameter received its first gradient after step 1 (unsupported)')
        for g in live:
            if g['buf'] is None:
                g['buf'] = torch.empty_like(g['p'], dtype=torch.float32)
        keep = [grad_of(g['p']).contiguous() for g in live]
        shadows = self.shadow_of() if self.shadow_of is not None else {}
        norm = self._norm.data_ptr() if self.clip and self.clip > 0 else None
        pending = []
        for dtype in (torch.float32, torch.bfloat16):          # gradients arrive fp32 (local) or bf16 (off the wire)
            idx = [i for i, gr in enumerate(keep) if gr.dtype == dtype]
            if not idx:
                continue
            arr = lambda vals: np.ascontiguousarray(np.array(vals, dtype=np.uint64))
            gp = arr([keep[i].data_ptr() for i in idx])
            pp = arr([live[i]['p'].data_ptr() for i in idx])
            bp = arr([live[i]['buf'].data_ptr() for i in idx])
            sh = [shadows.get(live[i]['name']) for i in idx]
            sp = arr([t.data_ptr() if t is not None else 0 for t in sh])
            nn = np.ascontiguousarray(np.array([keep[i].numel() for i in idx], dtype=np.int64))
            lr = np.ascontiguousarray(np.array([live[i]['lr'] for i in idx], dtype=np.float32))
            if norm is not None:
                _lib.call('sgg_sqnorm_multi', gp.ctypes.data, nn.ctypes.data, len(idx), norm, ops.dt(keep[idx[0]]), stream)
            pending.append((dtype, gp, pp, bp, sp, nn, lr, len(idx)))
        for dtype, gp, pp, bp, sp, nn, lr, cnt in pending:  # every norm contribution lands before the first update
            _lib.call('sgg_sgd_multi', pp.ctypes.data, gp.ctypes.data, bp.ctypes.data, sp.ctypes.data, nn.ctypes.data,
                      lr.ctypes.data, cnt, float(self.weight_decay), float(self.momentum), int(first), norm,
                      float(self.clip or 0.0), float(grad_scale), ops.dt(dtype), int(self.max_blocks), stream)
        self.wrote_shadow = [n for n in shadows if any(g['name'] == n for g in live)]
        self.steps += 1
        if self.on_update is not None:
            self.on_update()   # parameters changed through raw pointers: tell the owner to refresh derived operands

    def grad_norm(self):
        return float(self._norm.sqrt().item())


class Trainer(object):
    """One data-parallel train step.  `loss_type` as lib/losses.py ('baseline' is the reference default, config.py:184)."""

    def __init__(self, model, lr=1e-3, momentum=0.9, weight_decay=1e-4, clip=5.0, loss_type='baseline',
                 comm_dtype=torch.bfloat16, force_dist=False, sync_bn=True, pipeline=False):
        self.model = model
        for n, p in model.named_parameters():
            if n.startswith('detector.'):
                p.requires_grad = False                     # main.py:62-63: the detector is frozen
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        self.opt = FusedSGD(named, lr, momentum, weight_decay, clip)
        self.opt.on_update = self._bump
        if hasattr(model, 'shadow_buffers'):
            self.opt.shadow_of = model.shadow_buffers
        self.buckets = GradBuckets([p for _, p in named], comm_dtype=comm_dtype, force=force_dist)
        self.loss_type = loss_type
        # pipeline=True: the optimiser update of step k (HBM-bound, ~1.1 ms) and the rebuild of the weight-derived operands
        # run on the side stream UNDER the frozen VGG forward of step k+1 (MFMA-bound, ~2 ms), which does not read the
        # weights being updated (main.py:62-63: the detector is frozen).  The head of step k+1 waits for the event.
        # Every update still happens, in order; read parameters through flush() in this mode.
        self.pipeline = pipeline
        if pipeline:
            self.opt.max_blocks = 256      # leave wave slots for the VGG forward running beside the update
        self._queued = False
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # force_dist: run the distributed code path on a 1-rank group too (tests exercise RCCL plumbing on one GPU)
        self.dist_on = self.world > 1 or (force_dist and dist.is_available() and dist.is_initialized())
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
        if self.loss_type == 'baseline':
            if self.dist_on:
                # global normalisers stay on the device (fill_ = a launch, no host round trip): a .tolist() here would
                # stall the host between forward and backward on every rank
                t = torch.empty(2, dtype=torch.float32, device=res.rel_dists.device)
                t[0].fill_(n_obj)
                t[1].fill_(M)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                return obj_ce / t[0] + rel_ce.sum() / t[1]
            return obj_ce / n_obj + rel_ce.sum() / M                                 # lib/losses.py:41-42,74
        fg = labels > 0
        m_fg, m_bg = float(fg.sum().item()), float((~fg).sum().item())
        if self.world > 1:
            n_obj, m_fg, m_bg = sum_over_ranks([n_obj, m_fg, m_bg], device=res.rel_dists.device)
        w = torch.ones_like(rel_ce)
        if m_fg > 0:
            w[fg] = 1.0 / m_fg                                                        # :50
        if self.loss_type == 'dnorm':
            if m_bg > 0 and m_fg > 0:
                w[~fg] = 1.0 / m_fg                                                   # :56
        elif self.loss_type == 'dnorm-fgbg':
            if m_bg > 0:
                w[~fg] = 1.0 / m_bg                                                   # :59
        else:
            raise NotImplementedError(self.loss_type)
        return obj_ce / n_obj + (rel_ce * w).sum()

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
        """Make the current stream wait for a queued optimiser update (pipeline mode) before parameters are read."""
        ev = getattr(self.model, '_operands_ready', None)
        if ev is not None:
            torch.cuda.current_stream(next(self.model.parameters()).device).wait_event(ev)

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
            for g in self.opt.groups:                      # gradients were allocated on the main stream: keep their memory
                if g['p'].grad is not None:                # from being recycled there while the side stream reads it
                    g['p'].grad.record_stream(side)
            reduced = self.buckets.all_reduce(average=False) if self.dist_on else None
            for t in (reduced or {}).values():
                t.record_stream(side)
            self.opt.step(grads=reduced)
            train_weights(self.model)
            ev = torch.cuda.Event()
            ev.record(side)
        self.model._operands_ready = ev
        return True

    def step(self, batch):
        self.model.train()
        if not self._queued:
            self._prefetch_operands()
        res = self.model([batch])
        loss = self.losses(res)
        self.opt.zero_grad()
        loss.backward()
        self._queued = self.pipeline and self._queue_update()
        if not self._queued:
            reduced = self.buckets.all_reduce(average=False) if self.dist_on else None
            self.opt.step(grads=reduced)
        self.model.global_batch_iter = getattr(self.model, 'global_batch_iter', 0) + 1
        return loss.detach()
