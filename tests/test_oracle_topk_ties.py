"""The oracle's restatement of Tensor.topk's tie-breaking (libstdc++ nth_element / partial_sort on value-only pairs)
against torch.topk itself: the kept index SET must be identical on inputs built to tie at the k-th place."""
import numpy as np
import pytest
import torch

from oracle.topk_ties import topk_set_emulated


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
