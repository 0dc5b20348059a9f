"""The oracle's restatement of Tensor.topk's tie-breaking (libstdc++ nth_element / partial_sort on value-only pairs)
against torch.topk itself: the kept index SET must be identical on inputs built to tie at the k-th place, and the ORDER of
the returned indices too (its first entry is what util/util.py:159 drops: it decides the set when the best value is shared)."""
import numpy as np
import pytest
import torch

from oracle.topk_ties import topk_order_emulated, topk_set_emulated


@pytest.mark.parametrize("n,k", [(1024, 21), (768, 21), (1343, 21), (1344, 21), (2048, 21), (512, 41), (4096, 41), (100, 6)])
def test_emulated_topk_set_equals_torch(n, k):
    rs = np.random.RandomState(n + k)
    for trial in range(12):
        v = np.round(rs.randn(n) * rs.choice([2, 5, 20]), 0).astype(np.float32)     # coarse values: massive ties
        ref = sorted(torch.topk(torch.from_numpy(v), k).indices.tolist())
        assert topk_set_emulated(v, k) == ref, (n, k, trial)


def test_the_kept_set_is_not_the_lowest_index_rule():
    """Sanity: on these inputs 'ties -> lower index' is NOT what torch does, so the emulation is doing real work."""
    rs = np.random.RandomState(0)
    differs = 0
    for _ in range(20):
        v = np.round(rs.randn(1024) * 3, 0).astype(np.float32)
        ref = sorted(torch.topk(torch.from_numpy(v), 21).indices.tolist())
        low = sorted(np.lexsort((np.arange(1024), -v))[:21].tolist())
        differs += ref != low
    assert differs > 0


@pytest.mark.parametrize("n,k", [(1024, 21), (747, 21), (300, 21), (1343, 21), (1344, 21), (2048, 21), (512, 41), (2700, 41), (4096, 41),
                                 (100, 6), (100, 18), (333, 63), (4100, 63)])
def test_emulated_topk_order_equals_torch(n, k):
    """Every returned position, ties included -- coarse values (massive ties everywhere) and the kNN's own situation: distinct
    values below a best value that two, three or more than k entries share."""
    rs = np.random.RandomState(3 * n + k)
    for trial in range(12):
        if trial % 2 == 0:
            v = np.round(rs.randn(n) * rs.choice([2, 5, 20]), 0).astype(np.float32)
        else:
            v = -rs.rand(n).astype(np.float32)
            v[rs.permutation(n)[:rs.choice([2, 3, 5, k + 2])]] = 0.0
        ref = torch.topk(torch.from_numpy(v), k).indices.tolist()
        assert topk_order_emulated(v, k) == ref, (n, k, trial)


def test_rank0_among_shared_best_values_is_not_the_lowest_index():
    """Sanity: 'the lowest index of the tied best values comes first' is NOT what torch does -- the rule the HIP kernels had
    until round 6 (found by the vcrnetIter reuse soak: two launch forms kept different copies of a near-duplicate point)."""
    rs = np.random.RandomState(1)
    differs = 0
    for _ in range(40):
        v = -rs.rand(747).astype(np.float32)
        i, j = sorted(rs.permutation(747)[:2].tolist())
        v[i] = v[j] = 0.0
        differs += int(torch.topk(torch.from_numpy(v), 21).indices[0]) != i
    assert differs > 0
