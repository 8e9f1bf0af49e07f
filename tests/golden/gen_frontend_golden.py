"""Numeric pins of the frozen front end against the REFERENCE implementations.

* Swin-T: the reference's `SwinTransformer` (backbone/swin_transformer.py:501-760, built by
  `build_swin_transformer("swin_T_224_1k", 224, out_indices=(1,2,3), dilation=False)` as
  backbone.py:195-220 does), timm's DropPath / to_2tuple / trunc_normal_ stubbed (drop path is the
  identity in eval mode anyway).  Input 2 x 3 x 150 x 219: not a multiple of the patch size, of the
  window (7) or of the merge factor, so every padding branch runs.
* BERT: HuggingFace `BertModel` (transformers, bert-base-uncased geometry, random name-seeded weights,
  eval mode), with a padding mask -- the text encoder the reference wraps (bertwarper.py).

Weights are name-seeded (seeded.py): the fixture holds the inputs and the outputs only.
    python tests/golden/gen_frontend_golden.py      (needs /root/reference; never runs on the GPU box)
"""
import importlib
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
from seeded import fill_by_name_, layernorm_weights_plus_one_  # noqa: E402

SWIN_SALT, BERT_SALT = "frontend_swin/", "frontend_bert/"


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    ref_import.load()
    swin_mod = importlib.import_module("groundingdino.models.GroundingDINO.backbone.swin_transformer")
    misc = importlib.import_module("groundingdino.util.misc")
    swin = swin_mod.build_swin_transformer("swin_T_224_1k", 224, out_indices=(1, 2, 3), dilation=False)
    fill_by_name_(swin, SWIN_SALT, 0.05, {"norm": 0.1, "relative_position_bias_table": 0.5})
    layernorm_weights_plus_one_(swin)
    swin.eval()
    g = torch.Generator().manual_seed(5)
    img = torch.randn(2, 3, 150, 219, generator=g)
    mask = torch.zeros(2, 150, 219, dtype=torch.bool)
    mask[1, :, 180:] = True  # image 1 is narrower
    with torch.no_grad():
        outs = swin(misc.NestedTensor(img, mask))
    feats = [outs[k].tensors for k in sorted(outs)]
    fmasks = [outs[k].mask for k in sorted(outs)]

    from transformers import BertConfig, BertModel
    cfg = BertConfig()  # bert-base-uncased geometry
    bert = BertModel(cfg, add_pooling_layer=True)
    fill_by_name_(bert, BERT_SALT, 0.03, {"LayerNorm": 0.1, "embeddings": 0.2})
    with torch.no_grad():
        for name, p in bert.named_parameters():
            if "LayerNorm.weight" in name:
                p.add_(1.0)
    bert.eval()
    ids = torch.randint(1000, 20000, (2, 14), generator=g)
    ids[:, 0] = 101
    ids[0, 13] = 102
    ids[1, 9] = 102
    ids[1, 10:] = 0
    am = (ids != 0).long()
    tt = torch.zeros_like(ids)
    with torch.no_grad():
        hidden = bert(input_ids=ids, attention_mask=am, token_type_ids=tt)["last_hidden_state"]
    out = dict(swin_salt=SWIN_SALT, bert_salt=BERT_SALT,
               swin_param_names=[n for n, _ in swin.named_parameters()],
               bert_param_names=[n for n, _ in bert.named_parameters()],
               image=img, image_mask=mask, feats=feats, feat_masks=fmasks,
               input_ids=ids, attention_mask=am, token_type_ids=tt, last_hidden_state=hidden)
    path = os.path.join(HERE, "frontend.pt")
    torch.save(out, path)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1e3), [tuple(f.shape) for f in feats], tuple(hidden.shape),
          "feat std", [round(float(f.std()), 3) for f in feats], "hidden std %.3f" % float(hidden.std()))


if __name__ == "__main__":
    main()
