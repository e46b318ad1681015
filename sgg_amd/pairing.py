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


class PairedEdgeFeat(object):
    """The edge features as [U,C,P,P] pooled rows of the unordered pairs + the edge -> slot map; `dense()` is the reference's
    [E,C,P,P] tensor.  Behaves as that tensor wherever torch functions or tensor attributes are applied to it."""

    def __init__(self, rows, pairing):
        self.rows, self.pairing = rows, pairing
        self._dense = None

    def dense(self):
        if self._dense is None:
            self._dense = self.rows.index_select(0, self.pairing.e2u.long())
        return self._dense

    @property
    def shape(self):
        return torch.Size((self.pairing.E,) + tuple(self.rows.shape[1:]))

    def __getattr__(self, name):
        if name.startswith('__') or name in ('rows', 'pairing', '_dense'):     # (copy / pickle probe an object before __init__ ran)
            raise AttributeError(name)
        return getattr(self.dense(), name)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        def conv(a):
            if isinstance(a, PairedEdgeFeat):
                return a.dense()
            if isinstance(a, (tuple, list)):
                return type(a)(conv(x) for x in a)
            return a
        return func(*conv(tuple(args)), **{k: conv(v) for k, v in (kwargs or {}).items()})
