"""Unordered box pairs: the shared half of the edge branch.

The union box of (subject, object) and of (object, subject) is the same box, so the reference's per-edge RoIAlign
(sgg_models/rel_model_base.py:245-260) pools every unordered pair twice and `roi_fmap`'s first Linear (rel_model_stanford.py:104,
K = 25088) multiplies the same row twice; only `conv(rects)` (lib/get_union_boxes.py:99-101: which box is drawn in which channel)
tells the two directions apart, and it enters fc6 as a separate K = 512 term (DESIGN.md "fc6 fold").  With the boxes of an image
known on the host, slot u of pair {i < j} is a closed form, so the pooled rows and the long contraction are made once per slot
and every edge adds its own rect term -- the same numbers as the per-edge computation, for half of the two largest kernels'
work on complete graphs.  Used when the slots are at most 3/4 of the edges (sparse edge lists keep the per-edge path)."""
import numpy as np
import torch

from . import ops

_TABLES = {}


class EdgePairing(object):
    """pairs i64[U,2] (global box indices, i < j), e2u i32[E], u2e i32[U,2] (-1 = none), flag i32[1] (device; non-zero = the edge
    list broke the promise the host made: checked where a sync happens anyway)"""

    def __init__(self, pairs, e2u, u2e, flag, U, E):
        self.pairs, self.e2u, self.u2e, self.flag, self.U, self.E = pairs, e2u, u2e, flag, U, E


def _tables(counts, device):
    """per-image (first box, first slot, box count) i32[B] each and pairs i64[U,2], cached per tuple of box counts"""
    key = (tuple(counts), str(device))
    t = _TABLES.get(key)
    if t is None:
        first = np.concatenate(([0], np.cumsum(counts)[:-1])).astype(np.int32)
        per = np.array([n * (n - 1) // 2 for n in counts], dtype=np.int64)
        ubase = np.concatenate(([0], np.cumsum(per)[:-1])).astype(np.int32)
        rows = []
        for f, n in zip(first, counts):
            i, j = np.triu_indices(n, 1)                    # row-major: (0,1), (0,2), ..., (1,2), ... = slot order
            rows.append(np.stack((i + f, j + f), 1))
        pairs = np.concatenate(rows).astype(np.int64) if rows else np.zeros((0, 2), np.int64)
        t = (torch.from_numpy(first).to(device), torch.from_numpy(ubase).to(device),
             torch.from_numpy(np.asarray(counts, dtype=np.int32)).to(device), torch.from_numpy(pairs).to(device), int(per.sum()))
        if len(_TABLES) > 64:
            _TABLES.clear()
        _TABLES[key] = t
    return t


def make_pairing(rel_inds, segs, max_per_pair=2):
    """rel_inds i64[E,3] on the device, segs = [(image, first box, end box)] from the host -> EdgePairing, or None when the per-edge
    path is the better one (few edges per pair) or the host knows that a pair carries more than two edges."""
    E = rel_inds.shape[0]
    counts = [e - s for _, s, e in segs]
    if [i for i, _, _ in segs] != list(range(len(segs))) or (segs and segs[0][1] != 0) or max_per_pair > 2:
        return None
    U = sum(n * (n - 1) // 2 for n in counts)
    if U == 0 or 4 * U > 3 * E:
        return None
    first, ubase, cnt, pairs, U = _tables(counts, rel_inds.device)
    e2u, u2e, flag = ops.pair_slots(rel_inds.contiguous(), first, ubase, cnt, U)
    return EdgePairing(pairs, e2u, u2e, flag, U, E)


def make_pairing_symmetric(rel_inds, n_boxes, min_edges=256):
    """Pairing for a SYMMETRIC edge list -- (s, o) present iff (o, s) is -- sorted by (subject, object), e.g. the overlap-filtered
    pairs of get_rel_inds in eval mode (IoU is symmetric): the U = E / 2 edges with s < o are the slots, in list order; every other
    edge finds its partner by binary search.  Device-side torch index arithmetic, no host synchronisation: E is the only size needed.
    A list that breaks the promise sets `flag` (checked where the forward synchronises anyway)."""
    E = rel_inds.shape[0]
    if E < min_edges or E % 2:
        return None
    U = E // 2
    dev = rel_inds.device
    s, o = rel_inds[:, 1], rel_inds[:, 2]
    prim = s < o
    order = torch.sort((~prim).to(torch.uint8), stable=True).indices          # the edges with s < o first, both halves in list order
    prim_idx, rev_idx = order[:U], order[U:]
    keys = s * n_boxes + o                                                     # ascending along the list
    tgt = o[rev_idx] * n_boxes + s[rev_idx]
    pos = torch.searchsorted(keys, tgt).clamp_(max=E - 1)                      # the partner (o, s) of every edge with s > o
    slot_of = torch.cumsum(prim, 0, dtype=torch.int32) - 1                      # slot of an edge with s < o
    part = slot_of[pos].clamp_(0, U - 1)
    e2u = torch.empty(E, dtype=torch.int32, device=dev)
    e2u[prim_idx] = torch.arange(U, dtype=torch.int32, device=dev)
    e2u[rev_idx] = part
    u2e = torch.full((U, 2), -1, dtype=torch.int32, device=dev)
    u2e[:, 0] = prim_idx.to(torch.int32)
    u2e[part.long(), 1] = rev_idx.to(torch.int32)
    bad = (prim.sum() != U) | (keys[pos] != tgt).any() | (u2e[:, 1] < 0).any()
    return EdgePairing(rel_inds[prim_idx][:, 1:3].contiguous(), e2u, u2e, bad.to(torch.int32).view(1), U, E)


class PairedEdgeFeat(torch.Tensor):
    """The edge features as [U,C,P,P] pooled rows of the unordered pairs + the edge -> slot map.  A lazy `torch.Tensor`: it reports the
    shape / dtype / device of the reference's [E,C,P,P] tensor (`Result.edge_feat`, rel_model_base.py:245-260) without owning storage,
    and the first torch function, method, index or operator applied to it materialises that tensor (`dense()`, one row gather, cached)
    and carries on with it -- `len(res.edge_feat)`, `res.edge_feat[idx]`, arithmetic, `isinstance(..., torch.Tensor)` behave as they do
    on the reference's tensor (main.py:185 hands it to augment/gan.py:232-247).  `rows` / `pairing` are what this package's own
    predict() reads instead."""

    @staticmethod
    def __new__(cls, rows, pairing):
        t = torch.Tensor._make_wrapper_subclass(cls, (pairing.E,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device,
                                                requires_grad=False)
        t.rows, t.pairing, t._dense = rows, pairing, None
        return t

    def __init__(self, rows, pairing):
        pass

    def dense(self):
        if self._dense is None:
            self._dense = self.rows.index_select(0, self.pairing.e2u.long())
        return self._dense

    def __copy__(self):
        return PairedEdgeFeat(self.rows, self.pairing)

    def __deepcopy__(self, memo):
        import copy
        return PairedEdgeFeat(copy.deepcopy(self.rows, memo), copy.deepcopy(self.pairing, memo))

    def __reduce_ex__(self, proto):
        return (PairedEdgeFeat, (self.rows, self.pairing))

    def __repr__(self):
        return 'PairedEdgeFeat(%s rows of %d unordered pairs for %d edges, %s, %s)' % (
            tuple(self.rows.shape), self.pairing.U, self.pairing.E, self.rows.dtype, self.rows.device)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        # metadata of the wrapper itself (no reason to gather 100 MB to answer `.shape`)
        if func in _META and len(args) >= 1 and isinstance(args[0], PairedEdgeFeat):
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)

        def conv(a):
            if isinstance(a, PairedEdgeFeat):
                return a.dense()
            if isinstance(a, (tuple, list)):
                return type(a)(conv(x) for x in a)
            return a
        with torch._C.DisableTorchFunctionSubclass():
            return func(*conv(tuple(args)), **{k: conv(v) for k, v in kwargs.items()})

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        # reached only by callers that bypass __torch_function__ (C++ entry points): same rule, the dense tensor stands in
        from torch.utils._pytree import tree_map
        conv = lambda a: a.dense() if isinstance(a, PairedEdgeFeat) else a
        return func(*tree_map(conv, tuple(args)), **tree_map(conv, kwargs or {}))


_T = torch.Tensor
_META = {_T.shape.__get__, _T.dtype.__get__, _T.device.__get__, _T.ndim.__get__, _T.is_cuda.__get__, _T.requires_grad.__get__,
         _T.grad_fn.__get__, _T.is_leaf.__get__, _T.layout.__get__, _T.size, _T.dim, _T.numel, _T.nelement, _T.element_size,
         _T.__len__, _T.is_floating_point, _T.is_complex, _T.ndimension}
