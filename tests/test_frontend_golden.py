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
