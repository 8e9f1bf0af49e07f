"""ziragroundingdino_amd -- MI355X-native hot path of GroundingDINO / ZiRa.

Public surface mirrors the reference (JarintotionDin/ZiRaGroundingDINO):
``_C.ms_deform_attn_forward/backward``, ``MultiScaleDeformableAttnFunction``,
``MultiScaleDeformableAttention``, ``multi_scale_deformable_attn_pytorch`` (CPU tensors).  Importing the package does not load the HIP library;
the first op call does, and raises if it has not been built.
"""
__version__ = "0.1.0"

from . import _C  # noqa: F401
from .ms_deform_attn import (  # noqa: F401
    MultiScaleDeformableAttention,
    MultiScaleDeformableAttnFunction,
    multi_scale_deformable_attn_pytorch,
    sampling_locations_from_reference_points,
)
