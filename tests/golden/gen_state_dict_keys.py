"""EVERY state-dict key and shape of the reference's hot-path modules, for the checkpoint contract:
the reference `Transformer` with the GroundingDINO_SwinT_OGC_rep.py hyper-parameters and the heads its model file hangs on it
(groundingdino_dual_zero_rep_branch.py:321-361), its two Swin backbones (backbone/swin_transformer.py:775-780: swin_T_224_1k of
configs[1], swin_B_384_22k of configs[3]) and the reparameterizable side-branch modules (:57-135).  Data only: names and shapes.
    python tests/golden/gen_state_dict_keys.py      (needs /root/reference; never runs on the GPU box)
"""
import importlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
from gen_fullsize_golden import attach_heads, full_args  # noqa: E402


def shapes(module):
    return {k: list(v.shape) for k, v in module.state_dict().items()}


def main():
    ref = ref_import.load()
    tr = attach_heads(ref["transformer_for_adapter"].Transformer(**full_args()), ref["utils"].MLP, ref["utils"].ContrastiveEmbed)
    swin_mod = importlib.import_module("groundingdino.models.GroundingDINO.backbone.swin_transformer")
    zr = ref["groundingdino_dual_zero_rep_branch"]
    out = {
        "transformer": shapes(tr),
        "swin_T_224_1k": shapes(swin_mod.build_swin_transformer("swin_T_224_1k", 224, out_indices=(1, 2, 3), dilation=False)),
        "swin_B_384_22k": shapes(swin_mod.build_swin_transformer("swin_B_384_22k", 384, out_indices=(1, 2, 3), dilation=False)),
        "rep_zero_linear_768_256": shapes(zr.RepZeroLinear(768, 256)),
        "rep_zero_conv_192_256_1x1": shapes(zr.RepZeroConv2d(192, 256, kernel_size=1)),
        "rep_zero_conv_768_256_3x3s2": shapes(zr.RepZeroConv2d(768, 256, kernel_size=3, stride=2, padding=1)),
    }
    path = os.path.join(HERE, "state_dict_keys.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=0, sort_keys=True)
    print("wrote", path, {k: len(v) for k, v in out.items()}, "%.1f KB" % (os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    main()
