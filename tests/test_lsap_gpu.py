"""Device-side batched linear sum assignment (csrc/lsap.hip) against scipy.optimize.linear_sum_assignment --
what the reference's matcher calls per image (matcher.py:143-144) -- on the same float32 costs: the index
pairs must be IDENTICAL, also when many entries tie (scipy's tie-breaking is part of the contract)."""
import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment

pytestmark = pytest.mark.gpu

from ziragroundingdino_amd.lsap import infeasible, linear_sum_assignment_batched  # noqa: E402


def _check(cost, sizes, global_targets=False):
    S, B, Q, T = cost.shape
    q_idx, t_idx = linear_sum_assignment_batched(cost.cuda(), sizes, global_targets=global_targets)
    q_idx, t_idx = q_idx.cpu().numpy(), t_idx.cpu().numpy()
    toff = np.concatenate([[0], np.cumsum(sizes)])
    moff = np.concatenate([[0], np.cumsum([min(n, Q) for n in sizes])])
    assert q_idx.shape == (S, moff[-1])
    c = cost.numpy()
    for s in range(S):
        for b in range(B):
            rows, cols = linear_sum_assignment(c[s, b][:, toff[b]:toff[b + 1]])
            got_r, got_c = q_idx[s, moff[b]:moff[b + 1]], t_idx[s, moff[b]:moff[b + 1]]
            assert np.array_equal(got_r, rows), (s, b, got_r, rows)
            assert np.array_equal(got_c, cols + (toff[b] if global_targets else 0)), (s, b, got_c, cols)


@pytest.mark.parametrize("seed", range(4))
def test_random_costs_match_scipy(seed):
    g = torch.Generator().manual_seed(seed)
    sizes = [5, 0, 17, 64, 65][: 3 + seed % 3]
    _check(torch.randn(3, len(sizes), 50, sum(sizes), generator=g), sizes)


@pytest.mark.parametrize("levels", [1, 2, 3, 5])
def test_many_ties_match_scipy(levels):
    """Integer-valued costs from a handful of levels (1 level: a constant matrix): nearly every comparison
    inside the shortest-path search ties.  Covers wide (targets < queries), square and tall problems."""
    g = torch.Generator().manual_seed(levels)
    sizes = [7, 30, 45, 1]
    cost = torch.randint(0, levels, (4, len(sizes), 30, sum(sizes)), generator=g).float()
    _check(cost, sizes)
    _check(cost, sizes, global_targets=True)


def test_north_star_sizes_and_duplicate_predictions():
    """900 queries as in the model; duplicated queries and duplicated targets (exact ties in float costs)."""
    g = torch.Generator().manual_seed(7)
    sizes = [5, 40, 120]
    cost = torch.randn(7, 3, 900, sum(sizes), generator=g)
    cost[:, :, 450:] = cost[:, :, :450]            # every query appears twice
    cost[..., 10:20] = cost[..., 20:30]            # some targets too
    _check(cost, sizes, global_targets=True)


def test_state_in_global_memory_when_lds_is_too_small():
    g = torch.Generator().manual_seed(11)
    sizes = [2300, 3]
    _check(torch.rand(1, 2, 300, sum(sizes), generator=g), sizes)


def test_no_targets_and_infeasible_flag():
    q, t = linear_sum_assignment_batched(torch.zeros(2, 2, 10, 0).cuda(), [0, 0])
    assert q.shape == (2, 0) and t.shape == (2, 0)
    assert not infeasible(torch.device("cuda", torch.cuda.current_device()))
    bad = torch.rand(1, 1, 6, 3)
    bad[0, 0, :, 1] = float("inf")                # scipy: "cost matrix is infeasible"
    with pytest.raises(ValueError):
        linear_sum_assignment(bad[0, 0].numpy())
    linear_sum_assignment_batched(bad.cuda(), [3])
    assert infeasible(torch.device("cuda", torch.cuda.current_device()), reset=True)
    assert not infeasible(torch.device("cuda", torch.cuda.current_device()))
