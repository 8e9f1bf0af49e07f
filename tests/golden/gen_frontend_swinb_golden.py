"""Numeric pin of GroundingDINO-B's backbone (BASELINE configs[3]) against the REFERENCE implementation.

The reference's `build_swin_transformer("swin_B_384_22k", 384, out_indices=(1, 2, 3), dilation=False)`
(backbone/swin_transformer.py:775-780: embed_dim 128, depths 2/2/18/2, heads 4/8/16/32, window 12), timm stubbed as in
gen_frontend_golden.py, eval mode.  Input 2 x 3 x 150 x 219 with a ragged padding mask: 38 x 55 tokens after the patch
embedding -- not a multiple of the 12-wide windows, so the pad / shift / mask branches of every stage run, and the
144-token windows are the ones the 7 x 7 kernel does not cover.  Weights are name-seeded (seeded.py): the fixture holds
the inputs and the three output maps only.
    python tests/golden/gen_frontend_swinb_golden.py      (needs /root/reference; never runs on the GPU box)
"""
import importlib
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
from seeded import fill_by_name_, layernorm_weights_plus_one_  # noqa: E402

SWIN_SALT = "frontend_swinb/"


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    ref_import.load()
    swin_mod = importlib.import_module("groundingdino.models.GroundingDINO.backbone.swin_transformer")
    misc = importlib.import_module("groundingdino.util.misc")
    swin = swin_mod.build_swin_transformer("swin_B_384_22k", 384, out_indices=(1, 2, 3), dilation=False)
    fill_by_name_(swin, SWIN_SALT, 0.04, {"norm": 0.1, "relative_position_bias_table": 0.5})
    layernorm_weights_plus_one_(swin)
    swin.eval()
    g = torch.Generator().manual_seed(11)
    img = torch.randn(2, 3, 150, 219, generator=g)
    mask = torch.zeros(2, 150, 219, dtype=torch.bool)
    mask[1, :, 170:] = True  # image 1 is narrower
    with torch.no_grad():
        outs = swin(misc.NestedTensor(img, mask))
    feats = [outs[k].tensors for k in sorted(outs)]
    fmasks = [outs[k].mask for k in sorted(outs)]
    out = dict(swin_salt=SWIN_SALT, swin_param_names=[n for n, _ in swin.named_parameters()],
               image=img, image_mask=mask, feats=feats, feat_masks=fmasks)
    path = os.path.join(HERE, "frontend_swinb.pt")
    torch.save(out, path)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1e3), [tuple(f.shape) for f in feats],
          "feat std", [round(float(f.std()), 3) for f in feats], "params", sum(p.numel() for p in swin.parameters()))


if __name__ == "__main__":
    main()
