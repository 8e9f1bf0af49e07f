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
    q, t = linear_sum_assignment_batched(bad.cuda(), [3])
    assert infeasible(torch.device("cuda", torch.cuda.current_device()), reset=True)
    # the flagged problem still yields indices a gather can use (they index torch.empty memory otherwise)
    assert q.tolist() == [[0, 1, 2]] and t.tolist() == [[0, 1, 2]]
    nan = torch.rand(2, 2, 5, 7)
    nan[1, 0] = float("nan")
    q, t = linear_sum_assignment_batched(nan.cuda(), [4, 3], global_targets=True)
    assert infeasible(torch.device("cuda", torch.cuda.current_device()), reset=True)
    assert int(q.min()) >= 0 and int(q.max()) < 5 and int(t.min()) >= 0 and int(t.max()) < 7
    assert not infeasible(torch.device("cuda", torch.cuda.current_device()))


def test_fused_matching_cost_equals_the_pytorch_chain():
    """zira_match_cost_f32 against HungarianMatcher.cost_matrix (the reference's chain of PyTorch ops,
    matcher.py:105-141): the same float32 numbers to within 2e-6 (the L1 part bit for bit) and identical assignments."""
    from ziragroundingdino_amd.lsap import bad_boxes, matching_cost
    from ziragroundingdino_amd.matcher import HungarianMatcher

    g = torch.Generator().manual_seed(3)
    N, C, T = 3 * 900, 256, 23
    logits = (torch.randn(N, C, generator=g) * 3).cuda()
    boxes = torch.cat([torch.rand(N, 2, generator=g) * 0.6 + 0.2, torch.rand(N, 2, generator=g) * 0.3 + 0.01], -1).cuda()
    tgt = [{"labels": torch.randint(0, C, (T,), generator=g).cuda(),
            "boxes": torch.cat([torch.rand(T, 2, generator=g) * 0.5 + 0.25, torch.rand(T, 2, generator=g) * 0.3 + 0.1], -1).cuda()}]
    m = HungarianMatcher(cost_class=2.0, cost_bbox=5.0, cost_giou=2.0)
    want = m.cost_matrix({"pred_logits": logits[None], "pred_boxes": boxes[None]}, tgt)[0]
    got = matching_cost(logits, boxes, tgt[0]["labels"], tgt[0]["boxes"], 2.0, 5.0, 2.0, m.alpha, m.gamma)
    torch.testing.assert_close(got, want, rtol=2e-6, atol=2e-6)   # (last-bit differences in exp / log / divide remain)
    for a, b in zip(linear_sum_assignment(got.view(3, 900, T)[1].cpu().numpy()), linear_sum_assignment(want.view(3, 900, T)[1].cpu().numpy())):
        assert np.array_equal(a, b)
    dev = torch.device("cuda", torch.cuda.current_device())
    assert not bad_boxes(dev)
    boxes[5, 2] = -0.1                      # negative width: x1 < x0
    matching_cost(logits, boxes, tgt[0]["labels"], tgt[0]["boxes"])
    assert bad_boxes(dev, reset=True) and not bad_boxes(dev)


def test_category_logits_kernel_equals_pytorch_path():
    """zira_cat_logits_{fwd,bwd}_f32 against the PyTorch formulation of recover_to_cls_logits (CPU tensors take it):
    stacked leading dims, images with different numbers of categories / tokens, a category without tokens, a category
    whose tokens are all -inf (padding), exact ties between two tokens of a category."""
    from ziragroundingdino_amd.utils import recover_to_cls_logits

    g = torch.Generator().manual_seed(5)
    R, B, Q, T = 3, 2, 5, 16
    logits = torch.randn(R, B, Q, T, generator=g)
    logits[..., 12:] = float("-inf")                         # beyond the caption: masked token logits
    logits[:, 0, :, 2] = logits[:, 0, :, 1]                  # a tie inside category 0 of image 0
    m0 = torch.zeros(4, 14, dtype=torch.bool)
    m0[0, 1:3] = True; m0[1, 4:7] = True; m0[3, 12:14] = True     # category 2: no tokens; category 3: only -inf tokens
    m1 = torch.zeros(2, 9, dtype=torch.bool)
    m1[0, 1] = True; m1[1, 3:8] = True
    masks = [m0, m1]
    go = torch.randn(R, B, Q, T, generator=g)
    want_in = logits.clone().requires_grad_(True)
    want = recover_to_cls_logits(want_in, masks, for_fill=-100.0)
    (want_g,) = torch.autograd.grad(want, want_in, go)
    got_in = logits.cuda().requires_grad_(True)
    got = recover_to_cls_logits(got_in, [m.cuda() for m in masks], for_fill=-100.0)
    (got_g,) = torch.autograd.grad(got, got_in, go.cuda())
    assert torch.equal(got.cpu(), want)
    torch.testing.assert_close(got_g.cpu(), want_g, rtol=0, atol=0)
