"""The text-enhancer layer with its Linear layers on the row GEMMs (ziragroundingdino_amd/text_layer.py) against the same
layer running its modules (itself pinned to the reference's golden vectors): values and input gradients."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from ziragroundingdino_amd import text_layer, transformer  # noqa: E402

DEV = "cuda"


def _layer(seed=0):
    torch.manual_seed(seed)
    lay = transformer.TransformerEncoderLayer(d_model=256, nhead=4, dim_feedforward=1024, dropout=0.0).to(DEV).train()
    for p in lay.parameters():
        p.requires_grad_(False)
    with torch.no_grad():
        for n in (lay.norm1, lay.norm2):
            n.weight.uniform_(0.5, 1.5)
            n.bias.normal_(0, 0.1)
        lay.self_attn.in_proj_bias.normal_(0, 0.1)
        lay.linear1.bias.normal_(0, 0.1)
    return lay


def _inputs(T, B, seed):
    g = torch.Generator().manual_seed(seed)
    src = torch.randn(T, B, 256, generator=g).to(DEV).requires_grad_(True)
    pos = torch.randn(T, B, 256, generator=g).to(DEV)
    mask = torch.ones(B, T, T, dtype=torch.bool)
    for b in range(B):
        step = 3 + b
        for i in range(0, T, step):
            mask[b, i:i + step, i:i + step] = False      # True = not allowed: sub-sentence blocks, different per image
    return src, pos, mask.to(DEV), torch.randn(T, B, 256, generator=g).to(DEV)


def _run(lay, src, pos, mask, go, native):
    before = transformer.TransformerEncoderLayer.native_projections
    transformer.TransformerEncoderLayer.native_projections = native
    try:
        out = lay(src, src_mask=mask, src_key_padding_mask=None, pos=pos)
        return out.detach(), torch.autograd.grad([out], [src], [go])[0]
    finally:
        transformer.TransformerEncoderLayer.native_projections = before


@pytest.mark.parametrize("T,B", [(32, 2), (9, 2), (195, 1), (256, 2)])
def test_native_projections_match_modules(T, B):
    lay = _layer(T)
    src, pos, mask, go = _inputs(T, B, T + B)
    assert text_layer.applies(lay, src, pos)
    got, want = _run(lay, src, pos, mask, go, True), _run(lay, src, pos, mask, go, False)
    for name, a, b in zip(("out", "grad"), got, want):
        err = float((a - b).abs().max() / b.abs().max())
        assert err < 2e-5, (name, err)


def test_weights_refreshed_in_place_and_declines():
    lay = _layer(1)
    src, pos, mask, go = _inputs(32, 2, 5)
    _run(lay, src, pos, mask, go, True)
    ptrs = [t.data_ptr() for t in lay._text_layer_wt[1]]
    with torch.no_grad():
        lay.linear2.weight.mul_(0.5)
        lay.self_attn.in_proj_weight.add_(0.01)
    got, want = _run(lay, src, pos, mask, go, True), _run(lay, src, pos, mask, go, False)
    assert [t.data_ptr() for t in lay._text_layer_wt[1]] == ptrs
    for a, b in zip(got, want):
        assert float((a - b).abs().max() / b.abs().max()) < 2e-5
    lay.linear1.weight.requires_grad_(True)                  # a trainable Linear: the modules
    assert not text_layer.applies(lay, src, pos)
    lay.linear1.weight.requires_grad_(False)
    assert not text_layer.applies(lay, src.cpu(), pos.cpu())
    lay.eval()
    assert text_layer.applies(lay, src, pos)
