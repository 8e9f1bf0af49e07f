"""Frozen front end against the reference implementations (tests/golden/gen_frontend_golden.py): this
package's Swin-T against the reference's SwinTransformer, its BERT against HuggingFace's BertModel,
weights rebuilt from parameter names (same names, same shapes: the checkpoint contract).  fp32, 1e-4
of the tensor scale on the CPU; the GPU runs (hipBLASLt GEMMs, the row LayerNorm kernel) 1e-3."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from seeded import fill_by_name_, layernorm_weights_plus_one_  # noqa: E402

from ziragroundingdino_amd import backbone as zb  # noqa: E402
from ziragroundingdino_amd import bert as zbert  # noqa: E402
from ziragroundingdino_amd.utils import NestedTensor  # noqa: E402

DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]


def close(a, b, tol, what):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max()) / scale
    assert err <= tol, "%s: max err %.3e (scaled by %.3g) > %.1e" % (what, err, scale, tol)


@pytest.fixture(scope="module")
def golden():
    return torch.load(os.path.join(HERE, "golden", "frontend.pt"), weights_only=False)


@pytest.mark.parametrize("dev", DEVICES)
def test_swin_t_matches_reference(golden, dev):
    g = golden
    swin = zb.SwinTransformer(embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window_size=7,
                              out_indices=(1, 2, 3))
    assert [n for n, _ in swin.named_parameters()] == g["swin_param_names"]
    fill_by_name_(swin, g["swin_salt"], 0.05, {"norm": 0.1, "relative_position_bias_table": 0.5})
    layernorm_weights_plus_one_(swin)
    swin.to(dev).eval()
    with torch.no_grad():
        outs = swin(NestedTensor(g["image"].to(dev), g["image_mask"].to(dev)))
    tol = 1e-4 if dev == "cpu" else 1e-3
    for i, k in enumerate(sorted(outs)):
        close(outs[k].tensors, g["feats"][i], tol, "feature map %d" % i)
        assert torch.equal(outs[k].mask.cpu(), g["feat_masks"][i])


@pytest.mark.parametrize("dev", DEVICES)
def test_bert_matches_huggingface(golden, dev):
    g = golden
    bert = zbert.BertModel(zbert.BertConfig())
    assert [n for n, _ in bert.named_parameters()] == g["bert_param_names"]
    fill_by_name_(bert, g["bert_salt"], 0.03, {"LayerNorm": 0.1, "embeddings": 0.2})
    with torch.no_grad():
        for name, p in bert.named_parameters():
            if "LayerNorm.weight" in name:
                p.add_(1.0)
    bert.to(dev).eval()
    with torch.no_grad():
        hidden = bert(input_ids=g["input_ids"].to(dev), attention_mask=g["attention_mask"].to(dev),
                      token_type_ids=g["token_type_ids"].to(dev))["last_hidden_state"]
    valid = g["attention_mask"].bool()
    tol = 1e-4 if dev == "cpu" else 1e-3
    # (the rows of padded tokens attend to nothing real in either implementation: compare the real tokens)
    close(hidden.cpu()[valid], g["last_hidden_state"][valid], tol, "last_hidden_state")


@pytest.fixture(scope="module")
def golden_b():
    return torch.load(os.path.join(HERE, "golden", "frontend_swinb.pt"), weights_only=False)


def _swin_b(g, dev):
    swin = zb.SwinTransformer(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in zb.SWIN_VARIANTS["swin_B_384_22k"].items()},
                              out_indices=(1, 2, 3))
    assert [n for n, _ in swin.named_parameters()] == g["swin_param_names"]
    fill_by_name_(swin, g["swin_salt"], 0.04, {"norm": 0.1, "relative_position_bias_table": 0.5})
    layernorm_weights_plus_one_(swin)
    return swin.to(dev).eval()


@pytest.mark.parametrize("dev", DEVICES)
def test_swin_b_matches_reference(golden_b, dev):
    """GroundingDINO-B's backbone (BASELINE configs[3]: swin_B_384_22k, 12 x 12 windows, reference
    backbone/swin_transformer.py:775-780) against the reference's own SwinTransformer: fp32 1e-4 (CPU) / 1e-3 (GPU: the
    144-token window kernel, hipBLASLt GEMMs, the row LayerNorm kernel)."""
    g = golden_b
    swin = _swin_b(g, dev)
    with torch.no_grad():
        outs = swin(NestedTensor(g["image"].to(dev), g["image_mask"].to(dev)))
    tol = 1e-4 if dev == "cpu" else 1e-3
    for i, k in enumerate(sorted(outs)):
        close(outs[k].tensors, g["feats"][i], tol, "Swin-B feature map %d" % i)
        assert torch.equal(outs[k].mask.cpu(), g["feat_masks"][i])


@pytest.mark.gpu
def test_swin_b_bf16_autocast_against_the_reference(golden_b):
    """configs[3] runs its GEMMs under bf16 autocast: the same backbone against the REFERENCE's fp32 feature maps (not
    against this package's own fp32 run).  24 blocks of bf16 GEMMs with fp32 residuals / norms: 3e-2 of the map's scale
    at most, 6e-3 in the root mean square."""
    g = golden_b
    swin = _swin_b(g, "cuda")
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        outs = swin(NestedTensor(g["image"].cuda(), g["image_mask"].cuda()))
    for i, k in enumerate(sorted(outs)):
        got, want = outs[k].tensors.float().cpu(), g["feats"][i]
        scale = max(1.0, float(want.abs().max()))
        assert float((got - want).abs().max()) / scale <= 3e-2, (i, float((got - want).abs().max()) / scale)
        assert float((got - want).pow(2).mean().sqrt()) / scale <= 6e-3, (i, float((got - want).pow(2).mean().sqrt()) / scale)
