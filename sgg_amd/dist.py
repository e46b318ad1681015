"""One-process-per-GPU data parallelism (torch.distributed: 'nccl' == RCCL over xGMI on ROCm, 'gloo' on CPU tests).

The scene-graph path shards by image: every graph is per-image (block-diagonal adjacency, sgg_models/
rel_model_base.py:148), the detector is frozen, so inference needs no data-path collective; training needs one
gradient all-reduce over the 247.75 M trainable parameters plus 3 integers for the batch-level loss normalisers
(lib/losses.py:34-63).  SURVEY.md 8(e).
"""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torch.distributed.run contract)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kw = {}
        if backend == 'nccl':
            torch.cuda.set_device(local)
            kw['device_id'] = torch.device('cuda', local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local, world


def shard_range(n_items, rank, world):
    """Contiguous image shard [lo, hi) of rank `rank` (rank r takes images [r*B,(r+1)*B) when world divides n)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(value, device='cpu'):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values, device='cpu'):
    """All-reduce (sum) of a few python numbers, e.g. the loss normalisers (M, M_FG, N_obj)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist()


class GradBuckets(object):
    """Gradient all-reduce for the DP step.  The payload is dominated by four tensors (fc6 x2: 2 x 411 MB fp32,
    fc7 x2: 2 x 67 MB); everything else is < 13 MB in total.  Tensors >= `big_bytes` are reduced one by one (no packing
    copies), the small ones are packed into one flat bucket.  xGMI is point-to-point, so a ring all-reduce is per-link
    bound: few, large messages keep RCCL on its bandwidth-optimal path, and `comm_dtype=bfloat16` halves the bytes on
    the links (the sum is still applied to fp32 master gradients).  All collectives are issued asynchronously and
    waited for together."""

    def __init__(self, params, big_bytes=8 << 20, comm_dtype=None, force=False, shard=False):
        self.params = [p for p in params if p.requires_grad]
        self.comm_dtype = comm_dtype
        self.big = [p for p in self.params if p.numel() * p.element_size() >= big_bytes]
        self.small = [p for p in self.params if p.numel() * p.element_size() < big_bytes]
        self.force = force    # also run on a 1-rank group (tests)
        self._early = {}      # param -> (wire buffer, work handle): collectives launched from inside the backward
        self._keepalive = []
        # shard=True: the big tensors are REDUCE-SCATTERED instead of all-reduced -- rank r receives the summed gradient of its
        # contiguous 1/world of every big tensor only, the optimiser updates that part of the fp32 master / momentum, and the
        # updated compute-dtype operand is all-gathered (FusedSGD.step).  Same bytes on the links as the all-reduce (a ring
        # all-reduce IS a reduce-scatter + an all-gather), 1/world of the optimiser's HBM traffic per rank.
        self.shard = shard
        # measurement hooks (bench.py): bytes handed to collectives by the last step, and -- when `timing` is a list -- one event pair
        # per step around the waits on the wire handles (how long the waiting stream stood still for the links)
        self.bytes_last = 0
        self._bytes = 0
        self.timing = None

    def is_big(self, p):
        return any(p is q for q in self.big)

    def shard_of(self, p):
        """(lo, hi) element range of `p` (flattened) this rank owns, or None when `p` is not sharded."""
        if not self.shard or not self.is_big(p) or not (dist.is_available() and dist.is_initialized()):
            return None
        world, rank = dist.get_world_size(), dist.get_rank()
        n = p.numel()
        if (world == 1 and not self.force) or n % (world * 8):   # equal parts whose starts stay 16-byte aligned in every dtype used
            return None                            # (force: a 1-rank group still goes through the collectives -- tests)
        return rank * (n // world), (rank + 1) * (n // world)

    def _reduce(self, p, buf):
        """-> (tensor the optimiser will read, work handle)"""
        rng = self.shard_of(p)
        self._bytes += buf.numel() * buf.element_size()
        if rng is None:
            return buf, dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
        out = torch.empty(rng[1] - rng[0], dtype=buf.dtype, device=buf.device)
        work = dist.reduce_scatter_tensor(out, buf.reshape(-1), op=dist.ReduceOp.SUM, async_op=True)
        self._keepalive.append(buf)                # the input must outlive the collective
        return out, work

    def start(self, p, grad):
        """Launch the all-reduce (reduce-scatter when sharded) of one big gradient as soon as the backward has produced it
        (overlaps the rest of the backward); `all_reduce` later waits for it instead of issuing it."""
        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not self.force):
            return
        buf = grad.to(self.comm_dtype) if (self.comm_dtype is not None and self.comm_dtype != grad.dtype) else grad
        self._early[p] = self._reduce(p, buf.contiguous())

    def all_reduce(self, average=True):
        """average=True: p.grad of every parameter holds the mean gradient afterwards (returns {}).
        average=False (the trainer: losses are already normalised by the global batch): returns {param: reduced buffer}
        -- wire-dtype buffers of the big tensors and views into the flat bucket for the small ones -- for the fused
        optimiser to consume in place; nothing is copied back into p.grad."""
        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not self.force):
            return {}
        world = dist.get_world_size()
        if average and self.shard:
            raise ValueError('GradBuckets: average=True writes whole gradients back into p.grad; a sharded reduction leaves parts')
        direct = {}
        works, bufs = [], []
        for p in self.big:
            if p in self._early:
                buf, work = self._early.pop(p)
                if average and p.grad is None:
                    p.grad = torch.zeros_like(p)
                bufs.append((p, buf))
                works.append(work)
                continue
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            g = p.grad
            buf = g.to(self.comm_dtype) if (self.comm_dtype is not None and self.comm_dtype != g.dtype) else g
            buf, work = self._reduce(p, buf.contiguous())
            bufs.append((p, buf))
            works.append(work)
        # small tensors: one flat fp32 bucket, every slot padded to 4 elements so that views into it stay 16-byte aligned
        pieces, offs, off = [], [], 0
        for p in self.small:
            n = p.numel()
            pieces.append(p.grad.reshape(-1).float() if p.grad is not None else p.new_zeros(n, dtype=torch.float32))
            offs.append(off)
            pad = (-n) % 4
            if pad:
                pieces.append(p.new_zeros(pad, dtype=torch.float32))
            off += n + pad
        flat = torch.cat(pieces) if pieces else None
        if flat is not None:
            self._bytes += flat.numel() * flat.element_size()
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))
        ev = None
        if self.timing is not None and flat is not None and flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w in works:
            w.wait()
        if ev is not None:
            ev[1].record()
            self.timing.append(ev)
        self.bytes_last, self._bytes = self._bytes, 0
        self._keepalive = []
        for p, buf in bufs:
            if buf is not p.grad:
                if average:
                    p.grad.copy_(buf)
                else:
                    direct[p] = buf          # consumed as is (wire dtype) by the fused optimiser: no copy back
            if average:
                p.grad.div_(world)
        if flat is not None:
            if average:
                flat.div_(world)
            for p, o in zip(self.small, offs):
                g = flat[o:o + p.numel()].view_as(p)
                if not average:
                    direct[p] = g            # the optimiser reads the bucket in place
                elif p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
        return direct
