"""Row LayerNorm kernel (csrc/layernorm.hip, C ABI zira_layernorm_fwd_f32) against F.layer_norm in float64,
its autograd wrapper against autograd of F.layer_norm, and the nn.LayerNorm drop-in."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from ziragroundingdino_amd import dense  # noqa: E402

DEV = "cuda"


@pytest.mark.parametrize("rows,C", [(44446, 256), (133600, 96), (33400, 192), (8400, 384), (8200, 768),
                                    (8192, 1024), (9001, 4), (8195, 100), (8200, 516)])
@pytest.mark.parametrize("affine", [True, False])
def test_forward_matches_float64(rows, C, affine):
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=g) * 3 + 1.5).to(DEV)
    w = torch.randn(C, generator=g).to(DEV) if affine else None
    b = torch.randn(C, generator=g).to(DEV) if affine else None
    assert dense.layer_norm_supported(x, (C,), w, b)
    got = dense.layer_norm(x, (C,), w, b, 1e-5)
    want = F.layer_norm(x.double(), (C,), None if w is None else w.double(), None if b is None else b.double(), 1e-5)
    assert got.shape == x.shape and got.dtype == torch.float32
    assert float((got.double() - want).abs().max()) < 2e-5          # fp32 rounding of a two-pass LayerNorm
    ref = F.layer_norm(x, (C,), w, b, 1e-5)                          # ATen's fp32 kernel: same order of magnitude
    assert float((got - ref).abs().max()) < 2e-5
    assert torch.equal(got, dense.layer_norm(x, (C,), w, b, 1e-5))   # no atomics: bit-stable


def test_three_dimensional_input_and_statistics():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 22223, 256, generator=g).to(DEV)
    ln = dense.LayerNorm(256).to(DEV)
    with torch.no_grad():
        ln.weight.copy_(torch.randn(256, generator=g)); ln.bias.copy_(torch.randn(256, generator=g))
    got = ln(x)
    want = torch.nn.LayerNorm.forward(ln, x)
    assert got.shape == x.shape
    assert float((got - want).abs().max()) < 2e-5
    assert set(ln.state_dict()) == {"weight", "bias"}


def test_gradients_match_autograd_of_layer_norm():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 3000, 256, generator=g).to(DEV).requires_grad_(True)
    w = torch.randn(256, generator=g).to(DEV).requires_grad_(True)
    b = torch.randn(256, generator=g).to(DEV).requires_grad_(True)
    go = torch.randn(3, 3000, 256, generator=g).to(DEV)
    got = torch.autograd.grad((dense.layer_norm(x, (256,), w, b) * go).sum(), [x, w, b])
    want = torch.autograd.grad((F.layer_norm(x, (256,), w, b) * go).sum(), [x, w, b])
    for a, e in zip(got, want):
        assert float((a - e).abs().max() / e.abs().max()) < 1e-5
    # frozen affine parameters (the ZiRa fine-tune): only the input gradient is produced
    got_x, = torch.autograd.grad((dense.layer_norm(x, (256,), w.detach(), b.detach()) * go).sum(), [x])
    assert float((got_x - want[0]).abs().max() / want[0].abs().max()) < 1e-5


def test_small_or_odd_inputs_stay_with_aten():
    x = torch.randn(900, 256, device=DEV)
    assert not dense.layer_norm_supported(x, (256,), None, None)
    assert torch.equal(dense.layer_norm(x, (256,)), F.layer_norm(x, (256,)))
    y = torch.randn(9000, 30, device=DEV)                           # C % 4 != 0
    assert not dense.layer_norm_supported(y, (30,), None, None)
    z = torch.randn(9000, 64, device=DEV, dtype=torch.bfloat16)
    assert not dense.layer_norm_supported(z, (64,), None, None)


def test_bf16_autocast_returns_fp32_like_aten():
    x = torch.randn(9000, 256, device=DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        got = dense.layer_norm(x, (256,))
        want = F.layer_norm(x, (256,))
    assert got.dtype == want.dtype == torch.float32
    assert float((got - want).abs().max()) < 2e-5


@pytest.mark.parametrize("rows,C", [(44446, 256), (8200, 96), (8200, 384), (8200, 768), (8192, 1024), (8195, 100)])
@pytest.mark.parametrize("affine", [True, False])
def test_input_gradient_kernel_matches_float64(rows, C, affine):
    g = torch.Generator().manual_seed(rows * 3 + C)
    x = (torch.randn(rows, C, generator=g) * 2 - 0.5).to(DEV).requires_grad_(True)
    w = torch.randn(C, generator=g).to(DEV) if affine else None       # frozen: no grad -> the row kernel
    b = torch.randn(C, generator=g).to(DEV) if affine else None
    go = torch.randn(rows, C, generator=g).to(DEV)
    got, = torch.autograd.grad((dense.layer_norm(x, (C,), w, b) * go).sum(), [x])
    xd = x.detach().double().requires_grad_(True)
    want, = torch.autograd.grad((F.layer_norm(xd, (C,), None if w is None else w.double(),
                                              None if b is None else b.double()) * go.double()).sum(), [xd])
    assert float((got.double() - want).abs().max() / want.abs().max()) < 2e-6
