"""Import the reference's hot-path modules in the build container.

Only the golden-vector generators use this (they need /root/reference and never run on the GPU
box).  Missing third-party packages (torchvision, timm, detectron2, addict, yapf) are replaced by
empty stub modules; the CUDA-only extension ``groundingdino._C`` by an empty module, so the
reference's own pure-PyTorch CPU path is what gets evaluated.
"""
import importlib
import sys
import types

import torch

REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _box_area(boxes):
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


class _IdentityDropPath(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, x):
        return x


def load():
    """-> dict of the reference modules on the hot path."""
    import transformers  # noqa: F401  (must come before the torchvision stub)

    if REF not in sys.path:
        sys.path.insert(0, REF)
    import groundingdino

    groundingdino._C = _stub("groundingdino._C")
    tv = _stub("torchvision", __version__="0.19.0", _is_tracing=lambda: False)
    tv.ops = _stub("torchvision.ops")
    tv.ops.boxes = _stub("torchvision.ops.boxes", box_area=_box_area, nms=None)
    tv.models = _stub("torchvision.models")
    _stub("torchvision.models._utils", IntermediateLayerGetter=object)
    _stub("timm")
    _stub("timm.models")
    _stub("timm.models.layers", DropPath=_IdentityDropPath, to_2tuple=lambda x: (x, x),
          trunc_normal_=torch.nn.init.trunc_normal_)
    _stub("addict", Dict=dict)
    _stub("yapf")
    _stub("yapf.yapflib")
    _stub("yapf.yapflib.yapf_api", FormatCode=None)
    _stub("detectron2")
    _stub("detectron2.modeling", detector_postprocess=None)
    _stub("detectron2.structures", Boxes=object, ImageList=object, Instances=object)
    _stub("groundingdino.util.visualizer", COCOVisualizer=None)
    for pkg, path in (("groundingdino.models", REF + "/groundingdino/models"),
                      ("groundingdino.models.GroundingDINO", REF + "/groundingdino/models/GroundingDINO")):
        m = types.ModuleType(pkg)
        m.__path__ = [path]
        sys.modules[pkg] = m
    base = "groundingdino.models.GroundingDINO."
    names = ("ms_deform_attn", "transformer_for_adapter", "groundingdino_dual_zero_rep_branch",
             "groundingdino_dual_zero_rep_multilayer_branch", "criterion", "utils", "bertwarper", "fuse_modules",
             "transformer_vanilla")
    return {n: importlib.import_module(base + n) for n in names}
