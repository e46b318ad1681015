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
    """Flat gradient buckets for the DP all-reduce.  Parameters are packed in REVERSE registration order (heads, GRU
    and unary layers finish their backward first, fc6 -- 83 % of the bytes -- last), `bucket_bytes` per bucket, so
    that early buckets overlap with the remaining backward GEMMs.  xGMI is point-to-point, a ring all-reduce is
    per-link bound: few, large buckets (default 128 MiB) keep RCCL on its bandwidth-optimal path."""

    def __init__(self, params, bucket_bytes=128 << 20, dtype=None):
        self.params = [p for p in params if p.requires_grad]
        self.dtype = dtype
        self.buckets, cur, size = [], [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * (torch.finfo(dtype).bits // 8 if dtype else p.element_size())
            if cur and size + nbytes > bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
            cur.append(p)
            size += nbytes
        if cur:
            self.buckets.append(cur)
        self._flat = [None] * len(self.buckets)

    def all_reduce(self, average=True):
        """Sum (or mean) p.grad over ranks, bucket by bucket, asynchronously; returns after all are complete."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        world = dist.get_world_size()
        works = []
        for i, bucket in enumerate(self.buckets):
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in bucket]
            flat = torch.cat([g.reshape(-1).to(self.dtype or g.dtype) for g in grads])
            self._flat[i] = flat
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True))
        for i, (bucket, w) in enumerate(zip(self.buckets, works)):
            w.wait()
            flat, off = self._flat[i], 0
            if average:
                flat.div_(world)
            for p in bucket:
                n = p.numel()
                g = flat[off:off + n].view_as(p).to(p.dtype)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                off += n
            self._flat[i] = None
