"""The one-launch HIP form of gen_sineembed_for_position (csrc/refpoints.hip) against the PyTorch op chain it replaces
(which tests/test_modules_golden.py pins to the reference): bit-identical."""
import pytest
import torch

from ziragroundingdino_amd import utils

pytestmark = pytest.mark.gpu


def _chains(monkeypatch, fn, *args):
    monkeypatch.setattr(utils, "NATIVE_REFPOINT_OPS", False)
    want = fn(*args)
    monkeypatch.setattr(utils, "NATIVE_REFPOINT_OPS", True)
    return want, fn(*args)


@pytest.mark.parametrize("n", [2, 4])
def test_sine_embed_bit_identical(monkeypatch, n):
    g = torch.Generator().manual_seed(n)
    pos = torch.rand(900, 2, n, generator=g).cuda()
    pos[0, 0] = 0.0
    pos[1, 1] = 1.0
    want, got = _chains(monkeypatch, utils.gen_sineembed_for_position, pos)
    assert got.shape == (900, 2, n * 128) and torch.equal(got, want)
    # a non-contiguous view and a tensor that takes part in autograd
    view = torch.rand(2, 900, n, generator=g).cuda().transpose(0, 1)
    want, got = _chains(monkeypatch, utils.gen_sineembed_for_position, view)
    assert torch.equal(got, want)
    leaf = pos.clone().requires_grad_()
    out = utils.gen_sineembed_for_position(leaf)
    assert out.requires_grad      # (the PyTorch chain: the native form is for detached boxes only)


def test_box_head_matches_the_op_chain():
    """sigmoid(delta + inverse_sigmoid(ref)) as one node (utils.box_head) against the ATen chain it replaces
    (groundingdino_dual_zero_rep_branch.py:563-569): value within 2 ulp of the sigmoid (ATen's log and libm's differ in the
    last bit), both gradients at 1e-6 of their largest entry, reference boxes on and outside the clamps included."""
    import torch
    from ziragroundingdino_amd import utils
    g = torch.Generator().manual_seed(5)
    delta = (torch.randn(6, 2, 900, 4, generator=g) * 2).cuda()
    ref = torch.rand(6, 2, 900, 4, generator=g).cuda()
    ref.view(-1)[:8] = torch.tensor([0.0, 1.0, -0.1, 1.2, 1e-3, 1 - 1e-3, 5e-4, 0.9996]).cuda()
    res = []
    for native in (False, True):
        d, r = delta.clone().requires_grad_(True), ref.clone().requires_grad_(True)
        out = utils.box_head(d, r) if native else (d + utils.inverse_sigmoid(r)).sigmoid()
        gd, gr = torch.autograd.grad((out * torch.linspace(-1, 2, out.numel(), device="cuda").view_as(out)).sum(), [d, r])
        res.append((out.detach(), gd, gr))
    (o0, d0, r0), (o1, d1, r1) = res
    assert (o1 - o0).abs().max().item() <= 3e-7
    assert (d1 - d0).abs().max().item() <= 1e-6 * d0.abs().max().item()
    assert (r1 - r0).abs().max().item() <= 1e-6 * r0.abs().max().item()
    assert torch.equal(r1 == 0, r0 == 0)
