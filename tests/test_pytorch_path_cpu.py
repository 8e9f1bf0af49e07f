"""The product's pure-PyTorch MSDA path (ziragroundingdino_amd.multi_scale_deformable_attn_pytorch,
the counterpart of reference ms_deform_attn.py:90-130) against the golden vectors produced by the
reference's own function, including BASELINE configs[0] (B=1, Q=100, 1 level, 4 heads), and the
module taking that path for CPU tensors like the reference module does (:326-348).

Tolerances: fp32 2e-5 / fp64 1e-12 -- both sides are grid_sample evaluations of the same formula.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_msda_cases, load_npz
from ziragroundingdino_amd import MultiScaleDeformableAttention, multi_scale_deformable_attn_pytorch

TOL = {np.dtype("float32"): dict(rtol=2e-5, atol=2e-5), np.dtype("float64"): dict(rtol=1e-12, atol=1e-12)}


@pytest.mark.parametrize("path", golden_msda_cases(), ids=lambda p: os.path.basename(p)[5:-4])
def test_pytorch_path_matches_reference_golden(path):
    g = load_npz(path)
    tol = TOL[g["value"].dtype]
    t = lambda k: torch.from_numpy(g[k])
    value, loc, attn = t("value").requires_grad_(), t("sampling_loc").requires_grad_(), t("attn_weight").requires_grad_()
    out = multi_scale_deformable_attn_pytorch(value, t("spatial_shapes"), loc, attn)
    np.testing.assert_allclose(out.detach().numpy(), g["output"], **tol)
    out.backward(t("grad_output"))
    np.testing.assert_allclose(value.grad.numpy(), g["grad_value"], **tol)
    np.testing.assert_allclose(attn.grad.numpy(), g["grad_attn_weight"], **tol)
    scale = max(1.0, float(np.abs(g["grad_sampling_loc"]).max()))
    # (this path IS grid_sample + autograd, like the reference's: also equal on the pixel = -1 edge)
    np.testing.assert_allclose(loc.grad.numpy() / scale, g["grad_sampling_loc"] / scale, **tol)


def test_baseline_config0_is_covered():
    g = load_npz([p for p in golden_msda_cases() if "cfg0" in p][0])
    assert g["value"].shape[0] == 1 and g["sampling_loc"].shape[1] == 100
    assert g["spatial_shapes"].shape[0] == 1 and g["value"].shape[2] == 4


@pytest.mark.parametrize("refdim", [2, 4])
def test_module_runs_on_cpu_tensors(refdim):
    """MultiScaleDeformableAttention.forward on CPU tensors (the reference module's fallback branch,
    no native library involved) against the reference module's own output and gradients."""
    g = torch.load(os.path.join(GOLDEN, "mod_msda_module_ref%d.pt" % refdim), weights_only=False)
    mod = MultiScaleDeformableAttention(embed_dim=64, num_heads=4, num_levels=3, num_points=2, batch_first=True)
    mod.load_state_dict(g["state"])
    q = g["query"].clone().requires_grad_(True)
    v = g["value"].clone().requires_grad_(True)
    out = mod(query=q, value=v, query_pos=g["query_pos"], key_padding_mask=g["key_padding_mask"],
              reference_points=g["reference_points"], spatial_shapes=g["spatial_shapes"],
              level_start_index=g["level_start_index"])
    torch.testing.assert_close(out, g["out"], rtol=1e-4, atol=1e-4)
    params = dict(mod.named_parameters())
    grads = torch.autograd.grad(out, [q, v] + list(params.values()), g["grad_out"])
    torch.testing.assert_close(grads[0], g["grad_query"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(grads[1], g["grad_value"], rtol=1e-4, atol=1e-4)
    for (k, _), gr in zip(params.items(), grads[2:]):
        torch.testing.assert_close(gr, g["grad_params"][k], rtol=1e-4, atol=1e-4)


def test_level_tables_are_checked():
    """sum(H*W) and level_start_index are validated once per pair of level tensors."""
    g = torch.load(os.path.join(GOLDEN, "mod_msda_module_ref2.pt"), weights_only=False)
    mod = MultiScaleDeformableAttention(embed_dim=64, num_heads=4, num_levels=3, num_points=2, batch_first=True)
    bad = g["level_start_index"].clone()
    bad[1] += 1
    with pytest.raises(AssertionError, match="level_start_index"):
        mod(query=g["query"], value=g["value"], reference_points=g["reference_points"],
            spatial_shapes=g["spatial_shapes"], level_start_index=bad)
    with pytest.raises(AssertionError, match="cover"):
        mod(query=g["query"], value=g["value"][:, :-1], reference_points=g["reference_points"],
            spatial_shapes=g["spatial_shapes"], level_start_index=g["level_start_index"])


def test_native_nodes_decline_cpu_tensors():
    """The one-node forms of the decoder layer, the decoder glue and the encoder attention sublayer are GPU paths; on CPU
    tensors their ``applies()`` says no and the module composition runs (same outputs as the reference's modules)."""
    import types

    import torch

    from ziragroundingdino_amd import decoder_layer, encoder_layer, transformer
    from ziragroundingdino_amd.dense import LayerNorm
    from ziragroundingdino_amd.utils import MLP

    torch.manual_seed(0)
    dec = transformer.DeformableTransformerDecoderLayer(256, 2048, 0.0, "relu", 4, 8, 4, use_text_cross_attention=True)
    for p in dec.parameters():
        p.requires_grad_(False)
    tgt, pos = torch.randn(5, 2, 256), torch.randn(5, 2, 256)
    ref, text, value = torch.rand(5, 2, 4, 4), torch.randn(2, 3, 256), torch.randn(2, 30, 256)
    assert not decoder_layer.applies(dec, tgt, pos, ref, text, value, None, None)
    glue = types.SimpleNamespace(bbox_embed=torch.nn.ModuleList([MLP(256, 256, 4, 3)]), norm=LayerNorm(256),
                                 ref_point_head=MLP(512, 256, 256, 2), query_scale=None, query_pos_sine_scale=None)
    for m in (glue.bbox_embed, glue.norm, glue.ref_point_head):
        for p in m.parameters():
            p.requires_grad_(False)
    assert not decoder_layer.refine_applies(glue, 0, tgt, torch.rand(5, 2, 4))
    assert not decoder_layer.prep_applies(glue, torch.rand(5, 2, 4), torch.ones(2, 4, 2))
    enc = transformer.DeformableTransformerEncoderLayer(256, 1024, 0.0, "relu", 4, 8, 4)
    for p in enc.parameters():
        p.requires_grad_(False)
    src = torch.randn(2, 30, 256)
    shapes = torch.tensor([[4, 5], [2, 3], [1, 2], [1, 2]])
    assert not encoder_layer.applies(enc, src, torch.randn(2, 30, 256), torch.rand(2, 30, 4, 2), shapes, None)
