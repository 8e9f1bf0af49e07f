"""GroupNorm of the input projections (csrc/groupnorm.hip; reference groundingdino_dual_zero_rep_branch.py:487-529: nn.GroupNorm(32, 256)
behind each level's conv + side branch): forward and input gradient against torch's GroupNorm in float64 at the four level sizes of
the benchmark (H W not a multiple of 4 on two of them), with and without the second summand, beside ATen's fp32 result."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from ziragroundingdino_amd import dense  # noqa: E402


@pytest.mark.parametrize("B,C,H,W,G", [(2, 256, 100, 167, 32), (2, 256, 50, 84, 32), (2, 256, 25, 42, 32), (2, 256, 13, 21, 32),
                                       (1, 64, 3, 4, 8), (3, 96, 7, 9, 4)])
@pytest.mark.parametrize("two", [False, True])
def test_group_norm_matches_float64(B, C, H, W, G, two):
    torch.manual_seed(B * C + H)
    gn = torch.nn.GroupNorm(G, C).cuda()
    with torch.no_grad():
        gn.weight.uniform_(0.5, 1.5)
        gn.bias.normal_()
    for p in gn.parameters():
        p.requires_grad_(False)
    x = (torch.randn(B, C, H, W, device="cuda") * 3 + 5).requires_grad_(True)       # (a mean much larger than the spread)
    r = (torch.randn(B, C, H, W, device="cuda") * 0.1).requires_grad_(True) if two else None
    gy = torch.randn(B, C, H, W, device="cuda")
    assert dense.group_norm_supported(x, gn, r)
    y = dense.group_norm_frozen(x, gn, r)
    grads = torch.autograd.grad(y, [x] + ([r] if two else []), gy)
    gn64 = torch.nn.GroupNorm(G, C).cuda().double()
    gn64.load_state_dict({k: v.double() for k, v in gn.state_dict().items()})
    x64 = x.detach().double().requires_grad_(True)
    r64 = r.detach().double().requires_grad_(True) if two else None
    y64 = gn64(x64 + r64 if two else x64)
    g64 = torch.autograd.grad(y64, [x64] + ([r64] if two else []), gy.double())
    y32 = gn((x + r) if two else x)
    g32 = torch.autograd.grad(y32, [x], gy)[0]
    err = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
    assert err(y, y64) <= max(2 * err(y32, y64), 2e-6)
    assert err(grads[0], g64[0]) <= max(2 * err(g32, g64[0]), 2e-6)
    if two:
        assert torch.equal(grads[0], grads[1])
    assert torch.equal(y, dense.group_norm_frozen(x, gn, r))                       # fixed join order: bit-stable


def test_trainable_affine_and_other_layouts_fall_back():
    gn = torch.nn.GroupNorm(32, 256).cuda()
    x = torch.randn(2, 256, 8, 8, device="cuda")
    assert not dense.group_norm_supported(x, gn)                                     # affine parameters require grad
    for p in gn.parameters():
        p.requires_grad_(False)
    assert dense.group_norm_supported(x, gn)
    assert not dense.group_norm_supported(x.double(), gn)
    assert not dense.group_norm_supported(torch.randn(2, 256, 1, 3, device="cuda"), gn)   # H W < 4
