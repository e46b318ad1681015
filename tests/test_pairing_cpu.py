"""Host side of the unordered-pair edge branch (sgg_amd/pairing.py): slot tables and the lazy dense view."""
import copy

import numpy as np
import torch

from sgg_amd.pairing import EdgePairing, PairedEdgeFeat, _tables


def test_slot_tables_follow_the_closed_form():
    counts = [4, 1, 7, 2]
    first, ubase, cnt, pairs, U = _tables(counts, torch.device('cpu'))
    assert U == sum(n * (n - 1) // 2 for n in counts) == len(pairs)
    np.testing.assert_array_equal(first.numpy(), [0, 4, 5, 12])
    np.testing.assert_array_equal(cnt.numpy(), counts)
    f, ub = first.numpy(), ubase.numpy()
    for b, n in enumerate(counts):
        for i in range(n):
            for j in range(i + 1, n):
                u = ub[b] + i * (2 * n - i - 1) // 2 + (j - i - 1)        # the formula of pair_slots_kernel
                assert tuple(pairs[u].tolist()) == (f[b] + i, f[b] + j)
    assert _tables(counts, torch.device('cpu'))[3] is pairs                  # cached per box counts


def test_paired_edge_feat_behaves_like_the_dense_tensor():
    rows = torch.arange(3 * 2 * 2 * 2, dtype=torch.float32).view(3, 2, 2, 2)
    e2u = torch.tensor([0, 1, 0, 2, 2, 1], dtype=torch.int32)
    pm = EdgePairing(torch.zeros((3, 2), dtype=torch.int64), e2u, torch.zeros((3, 2), dtype=torch.int32), torch.zeros(1, dtype=torch.int32), 3, 6)
    ef = PairedEdgeFeat(rows, pm)
    dense = rows[e2u.long()]
    assert ef.shape == dense.shape and ef.dtype == dense.dtype
    assert torch.equal(ef.dense(), dense)
    assert torch.equal(ef.view(6, -1), dense.view(6, -1))                    # tensor attributes / methods
    assert torch.equal(torch.cat((ef, ef), 0), torch.cat((dense, dense), 0))  # torch functions
    assert torch.equal(torch.relu(torch.sub(ef, 5)), torch.relu(dense - 5))
    c = copy.copy(ef)                                                        # no attribute recursion before __init__
    assert torch.equal(c.dense(), dense)
    assert torch.equal(copy.deepcopy(ef).dense(), dense)


def test_paired_edge_feat_is_a_tensor_to_the_callers_of_result_edge_feat():
    """main.py:185 hands `res.edge_feat` to augment/gan.py:211-259, which takes len(), indexes it with a permutation and does arithmetic
    on the rows; isinstance(..., torch.Tensor) holds as for the reference's tensor.  Metadata does not materialise the dense tensor."""
    rows = torch.arange(3 * 2 * 2 * 2, dtype=torch.float32).view(3, 2, 2, 2)
    e2u = torch.tensor([0, 1, 0, 2, 2, 1], dtype=torch.int32)
    pm = EdgePairing(torch.zeros((3, 2), dtype=torch.int64), e2u, torch.zeros((3, 2), dtype=torch.int32), torch.zeros(1, dtype=torch.int32), 3, 6)
    ef = PairedEdgeFeat(rows, pm)
    dense = rows[e2u.long()]
    assert isinstance(ef, torch.Tensor)
    assert len(ef) == 6 and ef.dim() == 4 and ef.size(0) == 6 and ef.numel() == dense.numel() and ef.shape == dense.shape
    assert ef.dtype == torch.float32 and ef.device == rows.device and not ef.requires_grad and not ef.is_cuda
    assert ef._dense is None                                                 # none of the above gathered anything
    assert torch.equal(ef[0], dense[0]) and torch.equal(ef[torch.tensor([4, 1])], dense[[4, 1]]) and torch.equal(ef[1:3], dense[1:3])
    assert [tuple(r.shape) for r in ef] == [(2, 2, 2)] * 6                   # iteration
    for a, b in ((ef + 1, dense + 1), (1 + ef, 1 + dense), (ef * 2, dense * 2), (-ef, -dense), (ef / 2, dense / 2), (ef - ef, dense - dense),
                 (ef > 3, dense > 3), (ef == dense, dense == dense), (ef ** 2, dense ** 2)):
        assert type(a) is torch.Tensor and torch.equal(a, b)
    assert type(ef.to(torch.float64)) is torch.Tensor and ef.detach().shape == dense.shape and float(ef.mean()) == float(dense.mean())
    np.testing.assert_array_equal(np.asarray(ef), dense.numpy())
    np.testing.assert_array_equal(ef.cpu().numpy(), dense.numpy())

    def discriminator_loss_stand_in(features_real, features_fake):           # the access pattern of augment/gan.py:232-247
        n = len(features_real)
        idx = torch.arange(n - 1, -1, -1)[:len(features_fake)]
        real = features_real[idx].view(len(idx), -1)
        return (real - features_fake.view(len(idx), -1)).pow(2).mean()
    fake = torch.ones(4, 2, 2, 2)
    assert float(discriminator_loss_stand_in(ef, fake)) == float(discriminator_loss_stand_in(dense, fake))


def test_symmetric_sparse_lists():
    """make_pairing_symmetric: overlap-filtered style lists ((s,o) present iff (o,s)), sorted by (subject, object)"""
    from sgg_amd.pairing import make_pairing_symmetric
    rng = np.random.RandomState(3)
    counts = [9, 30, 4, 17]
    rows, first = [], 0
    for b, n in enumerate(counts):
        keep = np.triu(rng.rand(n, n) < 0.4, 1)
        keep = keep | keep.T
        for s in range(n):
            for o in range(n):
                if keep[s, o]:
                    rows.append((b, first + s, first + o))
        first += n
    rel = torch.tensor(rows, dtype=torch.int64)
    pm = make_pairing_symmetric(rel, first, min_edges=2)
    E = len(rows)
    assert pm is not None and pm.E == E and pm.U == E // 2 and int(pm.flag.item()) == 0
    pairs, e2u, u2e = pm.pairs.numpy(), pm.e2u.numpy(), pm.u2e.numpy()
    assert (pairs[:, 0] < pairs[:, 1]).all() and len({tuple(p) for p in pairs.tolist()}) == pm.U
    reln = rel.numpy()
    np.testing.assert_array_equal(pairs[e2u], np.stack((reln[:, 1:].min(1), reln[:, 1:].max(1)), 1))
    for u in range(pm.U):
        a, b = u2e[u]
        assert a != b and e2u[a] == u and e2u[b] == u and reln[a, 1] < reln[a, 2] and reln[b, 1] > reln[b, 2]
    # slots follow the list order of the edges with s < o
    assert (np.diff(u2e[:, 0]) > 0).all()
    # odd or short lists: per-edge path; a list that is not symmetric raises the flag (indices stay in range)
    assert make_pairing_symmetric(rel[:-1], first, min_edges=2) is None
    assert make_pairing_symmetric(rel[:10], first) is None
    broken = torch.cat((rel[:3], rel[5:], rel[-2:]))          # two edges lost their partners
    pb = make_pairing_symmetric(broken[:(len(broken) // 2) * 2], first, min_edges=2)
    assert pb is not None and int(pb.flag.item()) != 0
    assert int(pb.e2u.min()) >= 0 and int(pb.e2u.max()) < pb.U
