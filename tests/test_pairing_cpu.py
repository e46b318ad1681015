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
