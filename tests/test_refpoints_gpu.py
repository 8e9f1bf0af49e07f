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
