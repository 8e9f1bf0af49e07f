"""The encoder layer's deformable self-attention sublayer as one autograd node (encoder_layer.py) against the module
composition (reference transformer_for_adapter.py:888-899): same output, and the same gradient for the image tokens up to fp32
re-association and the samples whose location lies within an ulp of a pixel boundary (see test_decoder_layer_gpu)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach(), b.detach()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("B,shapes", [(2, [(25, 34), (13, 17), (7, 9), (4, 5)]), (1, [(16, 12), (8, 6), (4, 3), (2, 2)])])
@pytest.mark.parametrize("frozen_offsets", [True, False])
def test_attention_sublayer_matches_modules(B, shapes, frozen_offsets, monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from ziragroundingdino_amd import dense
    monkeypatch.setattr(dense, "LN_MIN_ROWS", 1)    # (the node takes the row LayerNorm kernel only where the modules do)
    from ziragroundingdino_amd import encoder_layer as native
    from ziragroundingdino_amd import transformer
    dev = torch.device("cuda")
    torch.manual_seed(0)
    layer = transformer.DeformableTransformerEncoderLayer(256, 2048, 0.0, "relu", 4, 8, 4).to(dev).train()
    with torch.no_grad():
        for name, p in layer.named_parameters():
            if "sampling_offsets" in name:
                continue    # (keep the module's initial pattern of offsets, perturbed below)
            p.normal_(0, 0.05) if p.dim() > 1 else p.normal_(0, 0.1)
        layer.norm1.weight.add_(1.0)
        layer.norm2.weight.add_(1.0)
        if not frozen_offsets:
            layer.self_attn.sampling_offsets.weight.normal_(0, 0.02)
    for p in layer.parameters():
        p.requires_grad_(False)
    S = sum(h * w for h, w in shapes)
    g = torch.Generator(device="cpu").manual_seed(1)
    src = torch.randn(B, S, 256, generator=g).to(dev).requires_grad_(True)
    pos = torch.randn(B, S, 256, generator=g).to(dev)
    sh = torch.tensor(shapes, device=dev)
    start = torch.cat([sh.new_zeros(1), (sh[:, 0] * sh[:, 1]).cumsum(0)[:-1]])
    ratios = torch.ones(B, 4, 2, device=dev)
    ref = transformer.TransformerEncoder.get_reference_points(shapes, ratios, device=dev)
    gout = torch.randn(B, S, 256, generator=g).to(dev)

    def run(on):
        layer.native_attention = on
        out = layer(src, pos, ref, sh, start, None)[0]
        (gs,) = torch.autograd.grad(out, [src], gout)
        return out, gs

    assert native.applies(layer, src, pos, ref, sh, None)
    calls = []
    orig = native.attention_sublayer
    native.attention_sublayer = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        got = run(True)
    finally:
        native.attention_sublayer = orig
    assert calls
    want = run(False)
    assert _rel(got[0], want[0]) < 2e-5
    if frozen_offsets:
        assert _rel(got[1], want[1]) < 2e-4, _rel(got[1], want[1])
    else:
        bad = (got[1] - want[1]).abs().amax(-1) > 2e-4 * float(want[1].abs().max())
        assert float(bad.float().mean()) < 0.02 and _rel(got[1], want[1]) < 0.1
    # declined: padding mask, trainable weight
    mask = torch.zeros(B, S, dtype=torch.bool, device=dev)
    assert not native.applies(layer, src, pos, ref, sh, mask)
    layer.self_attn.output_proj.weight.requires_grad_(True)
    assert not native.applies(layer, src, pos, ref, sh, None)
