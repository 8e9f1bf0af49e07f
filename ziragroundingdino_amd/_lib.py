"""ctypes binding of the C ABI declared in include/zira_msda.h.

There is deliberately no fallback: if ``libzira_msda.so`` is missing the import of the op
raises, so a GPU box can never silently run a non-HIP path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZIRA_MSDA_LIB") or os.path.join(_HERE, "libzira_msda.so")  # env: dev A/B builds

# every symbol include/zira_msda.h declares (tests check the .so exports exactly these)
SYMBOLS = (
    "zira_msda_fwd_f32", "zira_msda_bwd_f32", "zira_msda_fwd_f64", "zira_msda_bwd_f64",
    "zira_msda_bwd_workspace_bytes", "zira_msda_bwd_f32_ws",
    "zira_msda_plan_bytes", "zira_msda_plan_f32", "zira_msda_fwd_plan_f32", "zira_msda_bwd_planned_f32",
    "zira_msda_fwd_cpu_f32", "zira_msda_bwd_cpu_f32",
    "zira_rsb_workspace_floats", "zira_rsb_fwd_f32", "zira_rsb_bwd_f32",
    "zira_xty_workspace_floats", "zira_xty_f32",
    "zira_bisoftmax_workspace_floats", "zira_bisoftmax_fwd_f32", "zira_bisoftmax_bwd_f32",
    "zira_layernorm_fwd_f32", "zira_layernorm_bwd_f32", "zira_add_layernorm_fwd_f32",
    "zira_lsap_workspace_bytes", "zira_lsap_f32", "zira_match_cost_f32",
    "zira_cat_logits_fwd_f32", "zira_cat_logits_bwd_f32", "zira_window_attn_f32", "zira_window_attn_bf16",
    "zira_sine_embed_f32", "zira_attn_fwd_f32", "zira_attn_bwd_f32", "zira_attn_bwd_ld_f32", "zira_attn_bwd_scratch_floats", "zira_msda_sampling_fwd_f32", "zira_msda_sampling_bwd_f32", "zira_gemm_drelu_f32",
    "zira_rowgemm_f32", "zira_box_refine_fwd_f32", "zira_box_refine_bwd_f32", "zira_decoder_prep_f32",
    "zira_split_bf16x3_f32", "zira_gemm_bf16x3_f32", "zira_split_f16x2_f32", "zira_gemm_f16x2_f32", "zira_gemm_f16x2_ex_f32", "zira_split_f16x2_frag_f32", "zira_gemm_f16x2_panel_f32",
    "zira_ffn_f16x2_pack_bytes", "zira_ffn_f16x2_workspace_bytes", "zira_ffn_f16x2_pack_f32", "zira_ffn_f16x2_f32",
    "zira_thin_f16x2_frag_bytes", "zira_thin_f16x2_split_f32", "zira_thin_f16x2_f32",
    "zira_xty_bf16x3_workspace_floats", "zira_xty_bf16x3_f32",
    "zira_groupnorm_workspace_floats", "zira_groupnorm_fwd_f32", "zira_groupnorm_bwd_f32",
    "zira_stacked_losses_scratch_bytes", "zira_stacked_losses_fwd_f32", "zira_stacked_losses_bwd_f32",
    "zira_text_side_scratch_floats", "zira_text_prep_fwd_f32", "zira_text_prep_bwd_f32", "zira_text_out_fwd_f32", "zira_text_out_bwd_f32",
    "zira_sine_pos_hw_f32", "zira_box_head_fwd_f32", "zira_box_head_bwd_f32",
    "zira_level_valid_ratios_f32", "zira_encoder_ref_points_f32", "zira_encoder_proposals_f32",
    "zira_msda_version", "zira_msda_variant_f32",
)

_lib = None


class RowGemmArgs(ctypes.Structure):
    """``zira_rowgemm_args`` of include/zira_msda.h, field for field."""
    _fields_ = [
        ("a", ctypes.c_void_p), ("lda", ctypes.c_int),
        ("pos", ctypes.c_void_p), ("ldpos", ctypes.c_int), ("pos_cols", ctypes.c_int),
        ("w", ctypes.c_void_p), ("ldw", ctypes.c_int), ("w_is_nk", ctypes.c_int),
        ("bias", ctypes.c_void_p),
        ("res", ctypes.c_void_p), ("ldres", ctypes.c_int),
        ("mask", ctypes.c_void_p),
        ("relu", ctypes.c_int),
        ("ln_gamma", ctypes.c_void_p), ("ln_beta", ctypes.c_void_p), ("ln_eps", ctypes.c_float),
        ("ln_sum", ctypes.c_void_p), ("ln_mean", ctypes.c_void_p), ("ln_rstd", ctypes.c_void_p),
        ("lnb_x", ctypes.c_void_p), ("lnb_gamma", ctypes.c_void_p), ("lnb_mean", ctypes.c_void_p),
        ("lnb_rstd", ctypes.c_void_p), ("lnb_dx", ctypes.c_void_p),
        ("c", ctypes.c_void_p), ("ldc", ctypes.c_int),
        ("m", ctypes.c_int), ("n", ctypes.c_int), ("k", ctypes.c_int),
        ("batch", ctypes.c_int), ("a_batch_first", ctypes.c_int), ("c_batch_first", ctypes.c_int),
    ]


class ExtensionMissingError(ImportError):
    pass


def load():
    """Load (once) and return the ctypes handle of libzira_msda.so."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ExtensionMissingError(
            "HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or python -m ziragroundingdino_amd.build). There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    vp, i = ctypes.c_void_p, ctypes.c_int
    fwd_args = [vp, vp, vp, vp, vp, i, i, i, i, i, i, i, vp, vp]
    bwd_args = [vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, i, vp, vp, vp, vp]
    for suffix in ("f32", "f64"):
        f = getattr(lib, "zira_msda_fwd_" + suffix)
        f.argtypes, f.restype = fwd_args, i
        f = getattr(lib, "zira_msda_bwd_" + suffix)
        f.argtypes, f.restype = bwd_args, i
    f32 = ctypes.c_float
    lib.zira_attn_fwd_f32.argtypes = [vp] * 4 + [i] * 8 + [f32, vp, vp, vp]
    lib.zira_attn_fwd_f32.restype = i
    lib.zira_attn_bwd_f32.argtypes = [vp] * 7 + [i] * 8 + [f32] + [vp] * 4 + [ctypes.c_size_t, vp]
    lib.zira_attn_bwd_f32.restype = i
    lib.zira_attn_bwd_ld_f32.argtypes = [vp] * 7 + [i] * 8 + [f32] + [vp] * 3 + [i] * 3 + [vp, ctypes.c_size_t, vp]
    lib.zira_attn_bwd_ld_f32.restype = i
    lib.zira_attn_bwd_scratch_floats.argtypes = [i] * 4
    lib.zira_attn_bwd_scratch_floats.restype = ctypes.c_size_t
    lib.zira_msda_fwd_cpu_f32.argtypes, lib.zira_msda_fwd_cpu_f32.restype = fwd_args[:-1], i   # host pointers, no stream
    lib.zira_msda_bwd_cpu_f32.argtypes, lib.zira_msda_bwd_cpu_f32.restype = bwd_args[:-1], i
    lib.zira_msda_bwd_workspace_bytes.argtypes = [i] * 7
    lib.zira_msda_bwd_workspace_bytes.restype = ctypes.c_size_t
    lib.zira_msda_bwd_f32_ws.argtypes = bwd_args[:-1] + [vp, ctypes.c_size_t, vp]
    lib.zira_msda_bwd_f32_ws.restype = i
    lib.zira_msda_plan_bytes.argtypes = [i] * 7
    lib.zira_msda_plan_bytes.restype = ctypes.c_size_t
    lib.zira_msda_plan_f32.argtypes = [vp, vp, vp, vp] + [i] * 7 + [vp, ctypes.c_size_t, vp]
    lib.zira_msda_plan_f32.restype = i
    lib.zira_msda_fwd_plan_f32.argtypes = fwd_args[:-1] + [vp, ctypes.c_size_t, vp]
    lib.zira_msda_fwd_plan_f32.restype = i
    lib.zira_msda_bwd_planned_f32.argtypes = bwd_args[:-1] + [vp, ctypes.c_size_t, vp]
    lib.zira_msda_bwd_planned_f32.restype = i
    lib.zira_rowgemm_f32.argtypes = [ctypes.POINTER(RowGemmArgs), vp]
    lib.zira_decoder_prep_f32.argtypes = [vp, vp, vp, i, i, i, i, ctypes.c_float, vp, vp, vp, vp]
    lib.zira_decoder_prep_f32.restype = i
    lib.zira_box_refine_fwd_f32.argtypes = [vp, vp, vp, vp, ctypes.c_longlong, i, ctypes.c_float, vp, vp]
    lib.zira_box_refine_fwd_f32.restype = i
    lib.zira_box_refine_bwd_f32.argtypes = [vp, vp, vp, vp, ctypes.c_longlong, i, vp, vp]
    lib.zira_box_refine_bwd_f32.restype = i
    f32_ = ctypes.c_float
    lib.zira_stacked_losses_scratch_bytes.argtypes = [i, i, i, i]
    lib.zira_stacked_losses_scratch_bytes.restype = ctypes.c_size_t
    lib.zira_stacked_losses_fwd_f32.argtypes = [vp] * 8 + [i] * 5 + [f32_, f32_, vp, vp, vp]
    lib.zira_stacked_losses_fwd_f32.restype = i
    lib.zira_stacked_losses_bwd_f32.argtypes = [vp] * 9 + [i] * 5 + [f32_, f32_, vp, vp, vp]
    lib.zira_stacked_losses_bwd_f32.restype = i
    lib.zira_text_prep_fwd_f32.argtypes = [vp, vp, vp, f32_, vp, vp] + [i] * 5 + [vp] * 6
    lib.zira_text_prep_fwd_f32.restype = i
    lib.zira_text_side_scratch_floats.argtypes = [i] * 5
    lib.zira_text_side_scratch_floats.restype = ctypes.c_size_t
    lib.zira_text_prep_bwd_f32.argtypes = [vp] * 8 + [i] * 5 + [vp, vp, vp]
    lib.zira_text_prep_bwd_f32.restype = i
    lib.zira_text_out_fwd_f32.argtypes = [vp] * 7 + [i] * 5 + [vp, vp, vp]
    lib.zira_text_out_fwd_f32.restype = i
    lib.zira_text_out_bwd_f32.argtypes = [vp] * 6 + [i] * 5 + [vp, vp, vp]
    lib.zira_text_out_bwd_f32.restype = i
    lib.zira_sine_pos_hw_f32.argtypes = [vp, i, i, i, i, i, f32_, f32_, vp, vp, vp, vp]
    lib.zira_sine_pos_hw_f32.restype = i
    lib.zira_box_head_fwd_f32.argtypes = [vp, vp, ctypes.c_longlong, f32_, vp, vp]
    lib.zira_box_head_fwd_f32.restype = i
    lib.zira_box_head_bwd_f32.argtypes = [vp, vp, vp, ctypes.c_longlong, f32_, vp, vp, vp]
    lib.zira_box_head_bwd_f32.restype = i
    lib.zira_level_valid_ratios_f32.argtypes = [vp, vp, vp, i, ctypes.c_longlong, i, vp, vp, vp]
    lib.zira_level_valid_ratios_f32.restype = i
    lib.zira_encoder_ref_points_f32.argtypes = [vp, vp, vp, i, ctypes.c_longlong, i, vp, vp]
    lib.zira_encoder_ref_points_f32.restype = i
    lib.zira_encoder_proposals_f32.argtypes = [vp, vp, vp, i, ctypes.c_longlong, i, vp, vp, vp, vp]
    lib.zira_encoder_proposals_f32.restype = i
    lib.zira_rowgemm_f32.restype = i
    sz = ctypes.c_size_t
    lib.zira_rsb_workspace_floats.argtypes = [sz]
    lib.zira_rsb_workspace_floats.restype = sz
    lib.zira_rsb_fwd_f32.argtypes = [vp, vp, vp, sz, vp, vp, vp, vp]
    lib.zira_rsb_fwd_f32.restype = i
    lib.zira_rsb_bwd_f32.argtypes = [vp, vp, vp, vp, vp, sz, vp, vp, vp, vp, vp]
    lib.zira_rsb_bwd_f32.restype = i
    lib.zira_xty_workspace_floats.argtypes = [i, i, i, i]
    lib.zira_xty_workspace_floats.restype = sz
    lib.zira_xty_f32.argtypes = [vp, vp, i, i, i, i, i, vp, vp, vp]
    lib.zira_xty_f32.restype = i
    lib.zira_bisoftmax_workspace_floats.argtypes = [i, i, i, i]
    lib.zira_bisoftmax_workspace_floats.restype = sz
    lib.zira_bisoftmax_fwd_f32.argtypes = [vp, vp, vp, vp, i, i, i, i, i, i, i, vp, vp, vp, vp, vp, vp, vp]
    lib.zira_bisoftmax_fwd_f32.restype = i
    lib.zira_bisoftmax_bwd_f32.argtypes = [vp, vp, vp, i, i, i, i, i, i, i] + [vp] * 11
    lib.zira_bisoftmax_bwd_f32.restype = i
    lib.zira_layernorm_fwd_f32.argtypes = [vp, vp, vp, ctypes.c_int64, i, ctypes.c_float, vp, vp, vp, vp]
    lib.zira_layernorm_fwd_f32.restype = i
    lib.zira_layernorm_bwd_f32.argtypes = [vp, vp, vp, vp, vp, ctypes.c_int64, i, vp, vp]
    lib.zira_layernorm_bwd_f32.restype = i
    lib.zira_add_layernorm_fwd_f32.argtypes = [vp, vp, vp, vp, ctypes.c_int64, i, ctypes.c_float, vp, vp, vp, vp, vp]
    lib.zira_add_layernorm_fwd_f32.restype = i
    lib.zira_lsap_workspace_bytes.argtypes = [i, i, i, i]
    lib.zira_lsap_workspace_bytes.restype = sz
    lib.zira_lsap_f32.argtypes = [vp, i, i, i, i, i, vp, vp, vp, i, i, vp, vp, sz, vp]
    lib.zira_lsap_f32.restype = i
    f32 = ctypes.c_float
    lib.zira_match_cost_f32.argtypes = [vp, vp, vp, vp, i, i, i, f32, f32, f32, f32, f32, vp, vp, vp]
    lib.zira_match_cost_f32.restype = i
    ll = ctypes.c_longlong
    lib.zira_cat_logits_fwd_f32.argtypes = [vp, vp, vp, vp, ll, i, i, i, i, i, f32, vp, vp, vp]
    lib.zira_cat_logits_fwd_f32.restype = i
    lib.zira_cat_logits_bwd_f32.argtypes = [vp, vp, vp, ll, i, i, i, i, vp, vp]
    lib.zira_cat_logits_bwd_f32.restype = i
    lib.zira_window_attn_f32.argtypes = [vp, vp, vp, i, i, i, i, i, i, i, f32, vp, vp]
    lib.zira_window_attn_f32.restype = i
    lib.zira_window_attn_bf16.argtypes = [vp, vp, vp, i, i, i, i, i, i, i, f32, vp, vp]
    lib.zira_window_attn_bf16.restype = i
    lib.zira_msda_sampling_fwd_f32.argtypes = [vp, i, vp, i, vp, ll, i, i, i, vp, vp, vp]
    lib.zira_msda_sampling_fwd_f32.restype = i
    lib.zira_msda_sampling_bwd_f32.argtypes = [vp, vp, vp, vp, i, vp, ll, i, i, i, vp, i, vp]
    lib.zira_msda_sampling_bwd_f32.restype = i
    lib.zira_gemm_drelu_f32.argtypes = [vp, vp, vp, i, i, i, vp, vp]
    lib.zira_gemm_drelu_f32.restype = i
    lib.zira_split_bf16x3_f32.argtypes = [vp, i, i, i, vp, vp]
    lib.zira_split_bf16x3_f32.restype = i
    lib.zira_gemm_bf16x3_f32.argtypes = [vp, vp, i, i, i, i, vp, vp, vp, vp]
    lib.zira_gemm_bf16x3_f32.restype = i
    lib.zira_split_f16x2_f32.argtypes = [vp, i, i, i, vp, vp]
    lib.zira_split_f16x2_f32.restype = i
    lib.zira_gemm_f16x2_f32.argtypes = [vp, vp, i, i, i, i, vp, vp, vp, vp]
    lib.zira_gemm_f16x2_f32.restype = i
    lib.zira_gemm_f16x2_ex_f32.argtypes = [vp, vp, i, i, i, i, vp, vp, vp, i, vp, vp]
    lib.zira_gemm_f16x2_ex_f32.restype = i
    lib.zira_split_f16x2_frag_f32.argtypes = [vp, i, i, i, vp, vp]
    lib.zira_split_f16x2_frag_f32.restype = i
    lib.zira_gemm_f16x2_panel_f32.argtypes = [vp, vp, vp, i, i, i, i, vp, vp, vp, vp]
    lib.zira_gemm_f16x2_panel_f32.restype = i
    lib.zira_ffn_f16x2_pack_bytes.argtypes = [i]
    lib.zira_ffn_f16x2_pack_bytes.restype = ctypes.c_size_t
    lib.zira_ffn_f16x2_pack_f32.argtypes = [vp, ll, ll, vp, ll, ll, vp, i, vp, vp]
    lib.zira_ffn_f16x2_pack_f32.restype = i
    lib.zira_ffn_f16x2_workspace_bytes.argtypes = [i, i]
    lib.zira_ffn_f16x2_workspace_bytes.restype = ctypes.c_size_t
    lib.zira_ffn_f16x2_f32.argtypes = [vp, vp, i, i, i, vp, vp, vp, vp, vp, vp]
    lib.zira_ffn_f16x2_f32.restype = i
    lib.zira_groupnorm_workspace_floats.argtypes = [i, i, i, i]
    lib.zira_groupnorm_workspace_floats.restype = ctypes.c_size_t
    lib.zira_groupnorm_fwd_f32.argtypes = [vp, vp, vp, vp, i, i, i, i, f32, vp, vp, vp, vp, vp, vp]
    lib.zira_groupnorm_fwd_f32.restype = i
    lib.zira_groupnorm_bwd_f32.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, vp, vp, vp]
    lib.zira_groupnorm_bwd_f32.restype = i
    lib.zira_xty_bf16x3_workspace_floats.argtypes = [i, i, i]
    lib.zira_xty_bf16x3_workspace_floats.restype = ctypes.c_size_t
    lib.zira_xty_bf16x3_f32.argtypes = [vp, vp, i, i, i, i, vp, vp, vp]
    lib.zira_xty_bf16x3_f32.restype = i
    lib.zira_thin_f16x2_frag_bytes.argtypes = [i, i]
    lib.zira_thin_f16x2_frag_bytes.restype = ctypes.c_size_t
    lib.zira_thin_f16x2_split_f32.argtypes = [vp, i, i, i, i, vp, vp]
    lib.zira_thin_f16x2_split_f32.restype = i
    lib.zira_thin_f16x2_f32.argtypes = [vp, vp, vp, vp, i, i, i, i, vp, vp, vp, vp]
    lib.zira_thin_f16x2_f32.restype = i
    lib.zira_sine_embed_f32.argtypes = [vp, vp, ll, i, i, f32, vp, vp]
    lib.zira_sine_embed_f32.restype = i
    lib.zira_msda_version.restype = ctypes.c_char_p
    lib.zira_msda_variant_f32.argtypes = [i]
    lib.zira_msda_variant_f32.restype = ctypes.c_char_p
    _lib = lib
    return lib


def version():
    return load().zira_msda_version().decode()


def variant_f32(D):
    return load().zira_msda_variant_f32(int(D)).decode()
