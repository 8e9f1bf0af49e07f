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


@pytest.mark.parametrize("H,W,heads", [(25, 30, 4), (38, 55, 4), (12, 12, 2), (19, 28, 8)])
def test_12x12_windows_on_the_matrix_cores(H, W, heads):
    """The 144-token windows of the 384-pixel Swin-B / L variants (BASELINE configs[3]; reference
    backbone/swin_transformer.py:128-160, :222-270 with window_size 12) take the MFMA kernel: fp32 against the PyTorch
    formulation of the block (2e-5), and under bf16 autocast -- qkv and output bfloat16, the attention itself fp32 --
    against the same block's bf16 SDPA path (both round to bf16: 2e-2 of the scale) and its fp32 result (3e-2)."""
    torch.manual_seed(H * 100 + W + heads)
    dim = heads * 32
    layer = zb.BasicLayer(dim, 2, heads, 12, 4.0, [0.0, 0.0], downsample=False).cuda().eval()   # block 0 plain, 1 shifted
    for p in layer.parameters():
        torch.nn.init.normal_(p, std=0.1)
    x = torch.randn(2, H * W, dim, device="cuda")

    def run(native, bf16):
        zb.SwinTransformerBlock.native_attention = native
        try:
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
                return layer(x, H, W)[0].float()
        finally:
            zb.SwinTransformerBlock.native_attention = True

    want = run(False, False)
    got = run(True, False)
    scale = max(1.0, float(want.abs().max()))
    assert float((got - want).abs().max()) < 2e-5 * scale, float((got - want).abs().max())
    want_bf, got_bf = run(False, True), run(True, True)
    assert float((got_bf - want_bf).abs().max()) < 2e-2 * scale, float((got_bf - want_bf).abs().max())
    assert float((got_bf - want).abs().max()) < 3e-2 * scale, float((got_bf - want).abs().max())


def test_12x12_kernel_is_what_the_block_runs(monkeypatch):
    """configs[3] must not fall back to SDPA for its windows: count the native calls of one Swin-B-like layer, fp32 and
    under bf16 autocast."""
    from ziragroundingdino_amd import _lib
    lib = _lib.load()
    calls = {"f32": 0, "bf16": 0}
    real32, real16 = lib.zira_window_attn_f32, lib.zira_window_attn_bf16

    class Counted:
        def __init__(self, fn, key):
            self.fn, self.key = fn, key

        def __call__(self, *a):
            calls[self.key] += 1
            return self.fn(*a)

    monkeypatch.setattr(lib, "zira_window_attn_f32", Counted(real32, "f32"), raising=False)
    monkeypatch.setattr(lib, "zira_window_attn_bf16", Counted(real16, "bf16"), raising=False)
    layer = zb.BasicLayer(128, 2, 4, 12, 4.0, [0.0, 0.0], downsample=False).cuda().eval()
    x = torch.randn(1, 30 * 40, 128, device="cuda")
    with torch.no_grad():
        layer(x, 30, 40)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            layer(x, 30, 40)
    assert calls == {"f32": 2, "bf16": 2}, calls
