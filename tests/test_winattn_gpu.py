"""The native (shifted-)window attention of the frozen Swin blocks (csrc/winattn.hip) against the PyTorch formulation
of the same block (pad, roll, window partition, attention with bias and shift mask, reverse, roll back, crop --
reference backbone/swin_transformer.py:128-160, :222-270), which tests/test_frontend_golden.py pins to the reference."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from ziragroundingdino_amd import backbone as zb  # noqa: E402


@pytest.mark.parametrize("H,W,ws,heads", [(14, 21, 7, 3), (20, 17, 7, 6), (7, 7, 7, 2), (25, 30, 12, 4), (50, 84, 7, 12)])
def test_swin_layer_native_attention_equals_pytorch_path(H, W, ws, heads):
    torch.manual_seed(H * 100 + W)
    dim = heads * 32
    layer = zb.BasicLayer(dim, 2, heads, ws, 4.0, [0.0, 0.0], downsample=False).cuda().eval()   # block 0 plain, 1 shifted
    for p in layer.parameters():
        torch.nn.init.normal_(p, std=0.2)
    x = torch.randn(2, H * W, dim, device="cuda")
    with torch.no_grad():
        zb.SwinTransformerBlock.native_attention = False
        try:
            want = layer(x, H, W)[0]
        finally:
            zb.SwinTransformerBlock.native_attention = True
        zb.SwinTransformerBlock.native_max_tokens = 256    # (the model keeps 12x12 windows on SDPA; the kernel handles them)
        try:
            got = layer(x, H, W)[0]
        finally:
            zb.SwinTransformerBlock.native_max_tokens = 64
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) < 2e-5 * max(scale, 1.0), float((got - want).abs().max())


def test_native_attention_is_skipped_when_gradients_are_needed():
    layer = zb.BasicLayer(64, 2, 2, 7, 4.0, [0.0, 0.0], downsample=False).cuda()
    x = torch.randn(1, 14 * 14, 64, device="cuda", requires_grad=True)
    out = layer(x, 14, 14)[0]
    out.sum().backward()                         # the PyTorch path: differentiable
    assert x.grad is not None and torch.isfinite(x.grad).all()
