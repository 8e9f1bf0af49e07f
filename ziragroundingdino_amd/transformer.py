"""Cross-modal deformable transformer of GroundingDINO (the one every reference model variant
uses: groundingdino/models/GroundingDINO/transformer_for_adapter.py), plus its two encoder
neighbours (fuse_modules.py ``BiAttentionBlock``, transformer_vanilla.py
``TransformerEncoderLayer``).

Module / parameter names follow the reference so that its checkpoints load unchanged
(SURVEY.md appendix A): ``encoder.layers.N.self_attn.*``, ``encoder.text_layers.N.*``,
``encoder.fusion_layers.N.*``, ``decoder.layers.N.{cross_attn,ca_text,self_attn,...}``,
``decoder.ref_point_head``, ``level_embed``, ``tgt_embed``, ``enc_output``, ``enc_output_norm``.
The ablation-only ``use_adapter`` (Adapter / MoE) branch is not part of the ZiRa path
(``use_adapter = False`` in config/GroundingDINO_SwinT_OGC_rep.py:56) and is not built.

Every multi-scale deformable attention call (6 encoder layers with Q = S, 6 decoder layers
with Q = 900) runs the gfx950 kernels through ``MultiScaleDeformableAttention``.
"""
import math
import weakref
from typing import Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from .dense import (LayerNorm, bi_softmax, bi_softmax_supported, fusion_image_side, fusion_image_side_supported, tall_reduce,
                    tall_reduce_nt, wide_matmul, wide_matmul_residual, wide_matmul_residual_supported)
from .ms_deform_attn import MultiScaleDeformableAttention as MSDeformAttn
from .ms_deform_attn import multi_value_projections
from .utils import (MLP, _get_activation_fn, _get_clones, gen_encoder_output_proposals,
                    gen_sineembed_for_position, get_sine_pos_embed, inverse_sigmoid)


class DropPath(nn.Module):
    """Stochastic depth per sample (timm.models.layers.DropPath semantics)."""

    def __init__(self, drop_prob: float = 0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * mask.div_(keep)


def _cascade_reduce(x: Tensor, op: str) -> Tensor:
    """Full max / sum of x as a cascade of short row reductions (<= 256 elements each, one
    block per output) instead of one multi-block reduction."""
    v = x.reshape(-1)
    fill = float("-inf") if op == "max" else 0.0
    while v.numel() > 256:
        pad = (-v.numel()) % 256
        if pad:
            v = F.pad(v, (0, pad), value=fill)
        v = v.view(-1, 256)
        v = v.amax(dim=-1) if op == "max" else v.sum(dim=-1)
    return v.amax() if op == "max" else v.sum()


_ADDITIVE_MASKS = []   # [(weakref to the bool mask, its version, dtype, additive mask)], newest last


def _additive_mask(mask: Tensor, dtype) -> Tensor:
    """0 / -inf float form of a boolean "not allowed" mask.  The six decoder layers (and the text layers of the
    encoder) pass the SAME mask tensor: it is converted once, keyed on the tensor object (not its address)."""
    version = mask._version if not mask.is_inference() else 0
    for ref, ver, dt, out in _ADDITIVE_MASKS:
        if ref() is mask and ver == version and dt == dtype:
            return out
    out = torch.zeros_like(mask, dtype=dtype).masked_fill_(mask, float("-inf"))
    _ADDITIVE_MASKS.append((weakref.ref(mask), version, dtype, out))
    del _ADDITIVE_MASKS[:-4]
    return out


def lean_mha(mha: nn.MultiheadAttention, query: Tensor, key: Tensor, value: Tensor,
             key_padding_mask: Optional[Tensor] = None, attn_mask: Optional[Tensor] = None) -> Tensor:
    """``mha(query, key, value, key_padding_mask=..., attn_mask=..., need_weights=False)[0]`` for the
    configurations this model uses ([L, B, E] inputs, packed in-projection, no bias_kv / zero-attn),
    without the ~40 Python-level checks and views of ``F.multi_head_attention_forward`` (0.23 ms of
    host time per call, 18 calls per step -- it only shows when the host is the slower side).  Same
    projections, same scaled-dot-product kernel.  Boolean masks follow nn.MultiheadAttention's
    convention: True = not allowed."""
    L, B, E = query.shape
    S = key.shape[0]
    H = mha.num_heads
    hd = E // H
    w, b = mha.in_proj_weight, mha.in_proj_bias
    if key is query:  # one GEMM for the two projections of the same input
        qk = F.linear(query, w[:2 * E], b[:2 * E])
        q, k = qk.split(E, dim=-1)    # (split: its backward is one cat; two slices are two zero-fills, two copies and an add)
        v = F.linear(value, w[2 * E:], b[2 * E:])
    elif key is value:  # (cross-attention to one memory: decoder -> text)
        q = F.linear(query, w[:E], b[:E])
        kv = F.linear(key, w[E:], b[E:])
        k, v = kv.split(E, dim=-1)
    else:
        q = F.linear(query, w[:E], b[:E])
        k = F.linear(key, w[E:2 * E], b[E:2 * E])
        v = F.linear(value, w[2 * E:], b[2 * E:])
    dropout_p = mha.dropout if mha.training else 0.0
    if (Switches.fused_attention and attn_mask is None and dropout_p == 0.0 and hd == 32 and query.is_cuda
            and query.dtype == torch.float32 and not torch.is_autocast_enabled("cuda")):
        # the decoder's two small attentions: one HIP launch forward, two backward, scores in registers (attention.py)
        from . import attention
        kpm = key_padding_mask
        if kpm is not None and kpm.dtype == torch.bool:
            kpm = _additive_mask(kpm, q.dtype)
        if attention.supported(q, k, v, H, kpm):
            return F.linear(attention.fused_attention(q, k, v, H, kpm), mha.out_proj.weight, mha.out_proj.bias)
    return F.linear(_mha_core(q, k, v, H, key_padding_mask, attn_mask, dropout_p), mha.out_proj.weight, mha.out_proj.bias)


def _mha_core(q: Tensor, k: Tensor, v: Tensor, H: int, key_padding_mask, attn_mask, dropout_p: float) -> Tensor:
    """The attention between the in- and the out-projection of ``lean_mha``: q [L, B, E], k / v [S, B, E] (any strides) ->
    [L, B, E]; masks as nn.MultiheadAttention takes them."""
    L, B, E = q.shape
    S = k.shape[0]
    hd = E // H
    q = q.reshape(L, B * H, hd).transpose(0, 1).view(B, H, L, hd)
    k = k.reshape(S, B * H, hd).transpose(0, 1).view(B, H, S, hd)
    v = v.reshape(S, B * H, hd).transpose(0, 1).view(B, H, S, hd)
    mask = None
    if attn_mask is not None:
        if attn_mask.dtype == torch.bool:
            attn_mask = torch.zeros_like(attn_mask, dtype=q.dtype).masked_fill_(attn_mask, float("-inf"))
        mask = attn_mask.view(B, H, L, S) if attn_mask.dim() == 3 else attn_mask.view(1, 1, L, S)
    if key_padding_mask is not None:
        kpm = key_padding_mask
        if kpm.dtype == torch.bool:
            kpm = _additive_mask(kpm, q.dtype)
        kpm = kpm.view(B, 1, 1, S)
        mask = kpm if mask is None else mask + kpm
    if Switches.small_attention and B * H * L * S <= SMALL_ATTENTION_SCORES:
        out = _attention_small(q, k, v, mask, dropout_p)
    else:
        out = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=dropout_p)
    return out.permute(2, 0, 1, 3).reshape(L, B, E)


SMALL_ATTENTION_SCORES = 1 << 25   # score elements (B*H*L*S) up to which the scores are simply materialised


def _attention_small(q, k, v, mask, dropout_p=0.0):
    """softmax(q k^T / sqrt(d) + mask) v with the scores in memory: two batched GEMMs and a softmax.
    Every attention of this model outside the backbone is small (<= 900 x 900 per head, 16 heads x
    images), and the fused SDPA kernel of this stack runs such a problem on a handful of workgroups:
    61-74 us forward and 130-140 us backward per call, whatever the size (18 calls per step); the three
    plain kernels take a few microseconds each.  q, k, v: [B, H, L|S, d] (any strides)."""
    B, H, L, d = q.shape
    S = k.shape[2]
    q3, k3, v3 = q.reshape(B * H, L, d), k.reshape(B * H, S, d), v.reshape(B * H, S, d)
    scale = 1.0 / math.sqrt(d)
    if mask is not None:
        scores = torch.baddbmm(mask.expand(B, H, L, S).reshape(B * H, L, S), q3, k3.transpose(1, 2), alpha=scale)
    else:   # (beta = 0: the first argument is ignored; saves the q * scale kernel and its backward)
        scores = torch.baddbmm(q3.new_zeros(1, 1, 1).expand(B * H, L, S), q3, k3.transpose(1, 2), beta=0, alpha=scale)
    p = scores.softmax(-1)
    if dropout_p > 0.0:
        p = F.dropout(p, dropout_p)
    return torch.bmm(p, v3).view(B, H, L, d)


class Switches:
    """Module-level implementation switches (True = the leaner equivalent path)."""
    lean_mha = True
    small_attention = True   # lean_mha: materialised scores instead of the fused SDPA kernel for small problems
    native_geometry = True   # valid ratios / encoder reference points / two-stage proposals from the padding mask in one launch each (geometry.py)
    fused_attention = True   # lean_mha: csrc/attn.hip for fp32 heads of width 32 without attention mask / dropout (the decoder's)
    fused_ffn_backward = True  # frozen FFNs: (gy @ W2) * (h > 0) in one native GEMM (csrc/gemm_drelu.hip) instead of mm + threshold_backward
    sort_for_topk = False    # select_queries: stable sort instead of torch.topk everywhere (developer switch)
    # arithmetic of the frozen FFN products on the image-token rows: "f32" = the library's fp32 GEMMs (+ csrc/gemm_drelu.hip);
    # "bf16x3" = fp32-accurate split-bf16 products on the bf16 matrix cores (csrc/gemm_bf16x3.hip, gemm_bf16x3.py);
    # "f16x2" = the frozen FFN as ONE launch per direction on the f16 matrix cores in fp32 accuracy (csrc/ffn_f16x2.hip,
    # ffn_f16x2.py: the [rows, d_ffn] activation stays on chip), and "bf16x3" for the other frozen products.  The default since
    # round 6: both are closer to an fp64 evaluation than the library's fp32 GEMMs (tests/test_ffn_f16x2_gpu.py,
    # tests/test_gemm_bf16x3_gpu.py hold that gate; bench.py measures it in its line), and the step is 6 ms shorter.
    gemm_arith = "f16x2"


def _mha(mha, query, key, value, key_padding_mask=None, attn_mask=None):
    """lean_mha where it applies, nn.MultiheadAttention otherwise."""
    if (Switches.lean_mha and mha._qkv_same_embed_dim and mha.bias_k is None and not mha.add_zero_attn
            and not mha.batch_first and mha.in_proj_bias is not None and query.dim() == 3):
        return lean_mha(mha, query, key, value, key_padding_mask, attn_mask)
    return mha(query, key, value, key_padding_mask=key_padding_mask, attn_mask=attn_mask, need_weights=False)[0]


class _LinearReLU(torch.autograd.Function):
    """relu(x @ W^T + b) with bias and ReLU in the GEMM epilogue (``torch._addmm_activation``); the
    op has no autograd formula of its own, so the backward is written out.  x [N, K], W [H, K]."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        h = torch._addmm_activation(bias, x, weight.t())
        ctx.save_for_backward(x, weight, h)
        return h

    @staticmethod
    def backward(ctx, gh):
        x, weight, h = ctx.saved_tensors
        g = torch.ops.aten.threshold_backward(gh, h, 0)
        gx = g @ weight if ctx.needs_input_grad[0] else None
        gw = g.t() @ x if ctx.needs_input_grad[1] else None
        gb = g.sum(0) if ctx.needs_input_grad[2] else None
        return gx, gw, gb


class _FFNSplit:
    """The bf16 planes of a frozen FFN's two weights in the two orientations its forward and backward read them in
    (gemm_bf16x3.SplitWeight: made once, refreshed in place when a parameter changes)."""

    def __init__(self):
        from .gemm_bf16x3 import SplitWeight
        self.w1, self.w2, self.w2t, self.w1t = SplitWeight(False), SplitWeight(False), SplitWeight(True), SplitWeight(True)

    def refresh(self, lin1, lin2):
        for sw, w in ((self.w1, lin1.weight), (self.w1t, lin1.weight), (self.w2, lin2.weight), (self.w2t, lin2.weight)):
            sw.planes(w)


class _FFNFused:
    """The packed weights of a frozen FFN for the fused f16x2 launches (ffn_f16x2.PackedFFN: both directions, refreshed in place)."""
    fused = True

    def __init__(self):
        from .ffn_f16x2 import PackedFFN
        self.packed = PackedFFN()

    def refresh(self, lin1, lin2):
        self.packed.refresh(lin1, lin2)


def _ffn_split(layer, x2):
    """The layer's packed / split weights when ``Switches.gemm_arith`` asks for them and the shapes allow them, else None:
    "f16x2" -> _FFNFused where d_model = 256 and d_ffn % 256 == 0 (else as "bf16x3"), "bf16x3" -> _FFNSplit."""
    if Switches.gemm_arith not in ("bf16x3", "f16x2"):
        return None
    from . import gemm_bf16x3 as g3
    lin1, lin2 = layer.linear1, layer.linear2
    if Switches.gemm_arith == "f16x2" and x2.shape[0] >= 1024 and lin1.bias is not None and lin2.bias is not None:
        from . import ffn_f16x2 as ff
        if ff.supported(x2, lin1.out_features) and lin2.out_features == ff.D_MODEL and lin2.in_features == lin1.out_features:
            sp = getattr(layer, "_ffn_fused_weights", None)
            if sp is None:
                sp = layer._ffn_fused_weights = _FFNFused()
            return sp
    if not (g3.supported(x2, lin1.out_features, lin1.in_features) and lin2.out_features % 128 == 0 and lin2.in_features % 32 == 0
            and x2.shape[0] >= 1024):
        return None
    sp = getattr(layer, "_ffn_split_weights", None)
    if sp is None:
        sp = layer._ffn_split_weights = _FFNSplit()
    return sp


class _FrozenFFN(torch.autograd.Function):
    """linear2(relu(linear1(x))) with FROZEN weights (every ZiRa task): bias + ReLU in the first GEMM's epilogue, and in the
    backward the product  gy @ W2  masked by  h > 0  in ONE native kernel (csrc/gemm_drelu.hip) -- the separate
    threshold_backward pass over the [rows, d_ffn] gradient (1.1 GB of traffic per encoder layer) does not exist.
    x [N, K], W1 [F, K], W2 [K2, F]."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, split=None):
        ctx.split = split
        if getattr(split, "fused", False):   # one launch; the activation's sign bits are what the backward needs
            from . import ffn_f16x2 as ff
            F_ = w1.shape[0]
            bits = ff.mask_like(x, F_)
            y = ff.run(x, split.packed.get(w1, b1, w2, False), F_, False, bits, q_bias=b2)
            ctx.save_for_backward(w1, w2, bits, b1)
            return y
        if split is not None:
            from . import gemm_bf16x3 as g3
            h = g3.gemm(x, split.w1.planes(w1), g3.EPI_BIAS_RELU, bias=b1)
            ctx.save_for_backward(w1, w2, h)
            return g3.gemm(h, split.w2.planes(w2), g3.EPI_BIAS, bias=b2)
        h = torch._addmm_activation(b1, x, w1.t())
        ctx.save_for_backward(w1, w2, h)
        return torch.addmm(b2, h, w2.t())

    @staticmethod
    def backward(ctx, gy):
        from . import _lib
        gy = gy.contiguous()
        if getattr(ctx.split, "fused", False):
            from . import ffn_f16x2 as ff
            w1, w2, bits, b1 = ctx.saved_tensors
            gx = ff.run(gy, ctx.split.packed.get(w1, b1, w2, True), w1.shape[0], True, bits)
            return gx, None, None, None, None, None
        w1, w2, h = ctx.saved_tensors
        if ctx.split is not None:
            from . import gemm_bf16x3 as g3
            g = g3.gemm(gy, ctx.split.w2t.planes(w2), g3.EPI_MASK, aux=h)
            gx = g3.gemm(g, ctx.split.w1t.planes(w1), g3.EPI_BIAS, bias=torch.zeros(w1.shape[1], device=g.device))
            return gx, None, None, None, None, None
        g = torch.empty_like(h)
        with torch.cuda.device(h.device):
            rc = _lib.load().zira_gemm_drelu_f32(gy.data_ptr(), w2.data_ptr(), h.data_ptr(), h.shape[0], h.shape[1], gy.shape[1],
                                                 g.data_ptr(), torch.cuda.current_stream(h.device).cuda_stream)
        if rc != 0:
            raise RuntimeError("zira_gemm_drelu_f32 failed with code %d" % rc)
        return g @ w1, None, None, None, None, None


class _FrozenFFNNorm(torch.autograd.Function):
    """norm(x + linear2(relu(linear1(x)))) with FROZEN FFN and LayerNorm weights: _FrozenFFN followed by the residual
    LayerNorm kernel (dense._AddLayerNorm), as one autograd node so that the backward can add the gradient that reaches x
    through the residual connection inside the last GEMM (``addmm`` with beta = 1) instead of in a pass of its own
    (45 MB per tensor at the encoder shape).  x [N, K]."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, ln_w, ln_b, eps, split=None):
        from . import _lib
        ctx.split = split
        if getattr(split, "fused", False):
            from . import ffn_f16x2 as ff
            h = ff.mask_like(x, w1.shape[0])     # (the sign bits stand for the activation)
            y = ff.run(x, split.packed.get(w1, b1, w2, False), w1.shape[0], False, h, q_bias=b2)
        elif split is not None:
            from . import gemm_bf16x3 as g3
            h = g3.gemm(x, split.w1.planes(w1), g3.EPI_BIAS_RELU, bias=b1)
            y = g3.gemm(h, split.w2.planes(w2), g3.EPI_BIAS, bias=b2)
        else:
            h = torch._addmm_activation(b1, x, w1.t())
            y = torch.addmm(b2, h, w2.t())
        rows, C = x.shape
        out, s = torch.empty_like(x), torch.empty_like(x)
        stats = torch.empty((2, rows), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            rc = _lib.load().zira_add_layernorm_fwd_f32(x.data_ptr(), y.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), rows, C,
                                                        float(eps), s.data_ptr(), out.data_ptr(), stats[0].data_ptr(),
                                                        stats[1].data_ptr(), torch.cuda.current_stream(x.device).cuda_stream)
        if rc != 0:
            raise RuntimeError("zira_add_layernorm_fwd_f32 failed with code %d" % rc)
        ctx.save_for_backward(w1, w2, h, s, ln_w, stats, b1)
        return out

    @staticmethod
    def backward(ctx, gout):
        from . import _lib
        w1, w2, h, s, ln_w, stats, b1 = ctx.saved_tensors
        lib = _lib.load()
        gout = gout.contiguous()
        rows, C = s.shape
        fused = getattr(ctx.split, "fused", False)
        gs = torch.empty_like(s)
        g = None if fused else torch.empty_like(h)
        with torch.cuda.device(s.device):
            st = torch.cuda.current_stream(s.device).cuda_stream
            rc = lib.zira_layernorm_bwd_f32(gout.data_ptr(), s.data_ptr(), ln_w.data_ptr(), stats[0].data_ptr(),
                                            stats[1].data_ptr(), rows, C, gs.data_ptr(), st)
            if rc == 0 and ctx.split is None:   # gs: the gradient of x + ffn(x), i.e. of the FFN output and of x through the residual connection
                rc = lib.zira_gemm_drelu_f32(gs.data_ptr(), w2.data_ptr(), h.data_ptr(), rows, h.shape[1], C, g.data_ptr(), st)
        if rc != 0:
            raise RuntimeError("frozen FFN + LayerNorm backward failed with code %d" % rc)
        if fused:   # gs += ((gs W2) * [h > 0]) W1 in one launch, in place
            from . import ffn_f16x2 as ff
            return (ff.run(gs, ctx.split.packed.get(w1, b1, w2, True), w1.shape[0], True, h, aux=gs, out=gs),) + (None,) * 8
        if ctx.split is not None:   # the same two products on the bf16 matrix cores: ReLU mask and the residual sum in their epilogues
            from . import gemm_bf16x3 as g3
            g3.gemm(gs, ctx.split.w2t.planes(w2), g3.EPI_MASK, aux=h, out=g)
            return (g3.gemm(g, ctx.split.w1t.planes(w1), g3.EPI_ADD, aux=gs, out=gs),) + (None,) * 8
        return (gs.addmm_(g, w1),) + (None,) * 8   # (in place: torch.addmm(gs, ...) copies gs into its result first)


def _frozen_ffn_norm_ok(x, lin1, lin2, norm):
    from .dense import layer_norm_supported
    return (_frozen_ffn_ok(x, lin1, lin2) and isinstance(norm, LayerNorm) and norm.fused and LayerNorm.fused_residual
            and LayerNorm.fused_backward and norm.weight is not None and norm.bias is not None
            and not norm.weight.requires_grad and not norm.bias.requires_grad and x.is_contiguous()
            and layer_norm_supported(x, tuple(norm.normalized_shape), norm.weight, norm.bias))


def _frozen_ffn_ok(x, lin1, lin2):
    """The fused backward needs frozen fp32 weights with biases, d_ffn a multiple of 128 and d_model a multiple of 16."""
    return (Switches.fused_ffn_backward and x.is_cuda and x.dtype == torch.float32 and lin1.bias is not None
            and lin2.bias is not None and lin1.weight.dtype == torch.float32 and lin2.weight.dtype == torch.float32
            and not any(p.requires_grad for p in (lin1.weight, lin1.bias, lin2.weight, lin2.bias))
            and lin1.weight.is_contiguous() and lin2.weight.is_contiguous()
            and lin1.out_features % 128 == 0 and lin2.out_features % 16 == 0 and lin2.in_features == lin1.out_features)


def _max_over_tokens(x: Tensor) -> Tensor:
    """max over dim 1 of x [B, N, T] (N = 22 k image tokens, T = a few text tokens), shape
    [B, 1, T], with ``torch.max(dim)``'s gradient routing.  One ``max`` over the strided long
    dimension takes 250 us here (16 * T outputs, one block each); two short ones take ~15 us."""
    B, N, T = x.shape
    if N <= 512:
        return x.max(dim=1, keepdim=True)[0]
    pad = (-N) % 256
    if pad:
        x = F.pad(x, (0, 0, 0, pad), value=float("-inf"))
    return x.view(B, -1, 256, T).max(dim=2)[0].max(dim=1, keepdim=True)[0]


class _SubtractGlobalMax(torch.autograd.Function):
    """x - x.max() with the gradient autograd would give it (g - onehot(argmax) * sum(g), ties
    shared evenly) -- the reference's ``attn_weights - attn_weights.max()`` (fuse_modules.py:169).
    Written out because PyTorch's full reductions over this 2 M-element tensor (``max()`` in the
    forward, ``sum()`` in the backward) go through its multi-block reduce with global scratch,
    which replays once from a hipGraph and then faults on ROCm 7.2; the cascades below are plain
    row reductions and give the same values (max exactly, the sum up to fp32 association)."""

    @staticmethod
    def forward(ctx, x):
        m = _cascade_reduce(x, "max")
        ctx.save_for_backward(x, m)
        return x - m

    @staticmethod
    def backward(ctx, g):
        x, m = ctx.saved_tensors
        mask = (x == m).to(g.dtype)
        return g - mask * (_cascade_reduce(g, "sum") / _cascade_reduce(mask, "sum"))


class BiMultiHeadAttention(nn.Module):
    """Image <-> text attention sharing one score matrix (reference fuse_modules.py:99-248)."""

    fused_softmax = True  # score post-processing in one HIP op where it applies (fp32, no dropout)

    def __init__(self, v_dim, l_dim, embed_dim, num_heads, dropout=0.1, cfg=None):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == embed_dim
        self.v_dim, self.l_dim = v_dim, l_dim
        self.scale = self.head_dim ** (-0.5)
        self.dropout = dropout
        self.v_proj = nn.Linear(v_dim, embed_dim)
        self.l_proj = nn.Linear(l_dim, embed_dim)
        self.values_v_proj = nn.Linear(v_dim, embed_dim)
        self.values_l_proj = nn.Linear(l_dim, embed_dim)
        self.out_v_proj = nn.Linear(embed_dim, v_dim)
        self.out_l_proj = nn.Linear(embed_dim, l_dim)
        self.reassociate = True  # see forward(); False = the reference's order of operations
        self.stable_softmax_2d = True
        self.clamp_min_for_underflow = True
        self.clamp_max_for_overflow = True
        self._reset_parameters()

    def _reset_parameters(self):
        for lin in (self.v_proj, self.l_proj, self.values_v_proj, self.values_l_proj,
                    self.out_v_proj, self.out_l_proj):
            nn.init.xavier_uniform_(lin.weight)
            lin.bias.data.fill_(0)

    def _clamp(self, x: Tensor) -> Tensor:
        """clamp(min=-50000) then clamp(max=50000) (reference fuse_modules.py:171-177) as one kernel
        when both are switched on -- same values, same gradient."""
        lo = -50000 if self.clamp_min_for_underflow else None
        hi = 50000 if self.clamp_max_for_overflow else None
        return x if lo is None and hi is None else torch.clamp(x, min=lo, max=hi)

    def _heads(self, t: Tensor, bsz: int):
        return t.view(bsz, -1, self.num_heads, self.head_dim).transpose(1, 2).reshape(
            bsz * self.num_heads, -1, self.head_dim)

    compose_text_side = True   # class-level switch for A/B runs

    def _composed_text_side(self, l):
        """The text side of the re-bracketed products goes through ``embed_dim`` = 1024 twice in a row -- ``l_proj`` then the
        query weights, ``values_l_proj`` then the output weights, the value weights then ``out_l_proj`` -- as one Linear on
        the B x T <= 512 text tokens and one batched product per head: GEMMs of a few MFLOP that take 15-25 us each, and as
        many again in the backward.  While all six Linears are frozen (every ZiRa task) each pair is ONE constant matrix:
            a_h  = scale (l Wl_h^T + bl_h) Wq_h         = l A_h + a0_h        A_h = scale Wl_h^T Wq_h        [l_dim, v_dim]
            c_h  = scale (l Wl_h^T + bl_h) . bq_h       = l C_h + c0_h
            z_h  = (l Wvl_h^T + bvl_h) Wo_h^T           = l Z_h + z0_h        Z_h = Wvl_h^T Wo_h^T
            out_l = sum_h (u_h Wvv_h^T + bvv_h) Wol_h^T + bol = [u_0 .. u_H] O + o0,   O_h = Wvv_h^T Wol_h^T
        (products formed once in float64, rounded to fp32; fp32 re-association as the image side, reference
        fuse_modules.py:152-163, :213-235).  Rebuilt -- IN PLACE, captured graphs keep reading the buffers -- whenever a
        weight changes (load_state_dict, .to(), an optimizer that does train them elsewhere).  Returns None when it does
        not apply."""
        lins = (self.l_proj, self.v_proj, self.values_l_proj, self.out_v_proj, self.values_v_proj, self.out_l_proj)
        if not self.compose_text_side or not l.is_cuda or l.dtype != torch.float32 or torch.is_autocast_enabled("cuda"):
            return None
        ps = [p for lin in lins for p in (lin.weight, lin.bias)]
        if any(p is None or p.requires_grad or p.dtype != torch.float32 or p.device != l.device for p in ps):
            return None
        key = tuple(x for p in ps for x in (p.data_ptr(), p._version))
        cached = getattr(self, "_text_side", None)
        if cached is None or cached[0] != key:
            H, hd = self.num_heads, self.head_dim
            with torch.no_grad():
                f64 = lambda t: t.detach().double()
                Wl, bl = f64(self.l_proj.weight).view(H, hd, -1), f64(self.l_proj.bias).view(H, hd)
                Wq, bq = f64(self.v_proj.weight).view(H, hd, -1), f64(self.v_proj.bias).view(H, hd)
                Wvl, bvl = f64(self.values_l_proj.weight).view(H, hd, -1), f64(self.values_l_proj.bias).view(H, hd)
                Wo = f64(self.out_v_proj.weight).view(-1, H, hd)
                Wvv, bvv = f64(self.values_v_proj.weight).view(H, hd, -1), f64(self.values_v_proj.bias).view(H, hd)
                Wol, bol = f64(self.out_l_proj.weight).view(-1, H, hd), f64(self.out_l_proj.bias)
                A = self.scale * torch.einsum("hel,hed->lhd", Wl, Wq).flatten(1)          # [l_dim, H * v_dim]
                C = self.scale * torch.einsum("hel,he->lh", Wl, bq)                       # [l_dim, H]
                a0 = self.scale * torch.einsum("he,hed->hd", bl, Wq).flatten()
                c0 = self.scale * (bl * bq).sum(-1)
                new = [torch.cat([A, C], 1), torch.cat([a0, c0]),
                       torch.einsum("hel,dhe->lhd", Wvl, Wo).flatten(1), torch.einsum("he,dhe->hd", bvl, Wo).flatten(),
                       torch.einsum("hed,lhe->hdl", Wvv, Wol).flatten(0, 1), torch.einsum("he,lhe->l", bvv, Wol) + bol]
                new = [t.float().contiguous() for t in new]
                w1 = torch.cat([new[0], new[2]], 1).contiguous()      # [AC | Z], its bias, and the two transposes text_side.py reads
                new += [w1, torch.cat([new[1], new[3]]).contiguous(), w1.t().contiguous(), new[4].t().contiguous()]
                if cached is not None and all(o.shape == n.shape and o.device == n.device for o, n in zip(cached[1], new)):
                    for o, n in zip(cached[1], new):
                        o.copy_(n)
                    new = cached[1]
            cached = self._text_side = (key, new)
        return cached[1]

    def refresh_fused_projection(self, *unused):
        """The composed text-side matrices follow the six Linears in place (GraphedTransformer calls this when a parameter
        changed: with ``graph_fusion`` a replayed block re-runs no Python, so the key check of ``_composed_text_side`` would
        never see a weight loaded into a live model)."""
        cached = getattr(self, "_text_side", None)
        if cached is not None:
            self._composed_text_side(cached[1][0])   # (any fp32 tensor on the right device re-runs the key check)

    def forward(self, v, l, attention_mask_v=None, attention_mask_l=None, residual_v=None):
        """v: image tokens [B, N, v_dim] (N = 22 k), l: text tokens [B, T, l_dim] (T <= 256).
        ``residual_v``: optional callable returning the scale of the caller's ``v + scale * out_v``; when the fused path can
        take the residual into its last GEMM it returns ``(v + scale * out_v, out_l, True)`` instead of ``(out_v, out_l)``.

        ``reassociate`` (default): the three image-side projections (v_dim -> embed_dim = 1024 on N
        tokens, 70 GFLOP and five 91 MB tensors per layer at the bench shape) are never formed.
        Because T is small, the products are re-bracketed around the text side:
            scores    (v Wq^T + bq) k^T        = v (Wq^T k^T) + bq k^T          [N x 256] @ [256 x H*T]
            text out  P_l (v Wv^T + bv)        = (P_l v) Wv^T + rowsum(P_l) bv  [H*T x N] @ [N x 256]
            image out concat_h(P_v value_l) Wo^T = [P_v]_h-cat (value_l Wo^T)   [N x H*T] @ [H*T x 256]
        -- the same numbers up to fp32 re-association (checked against the reference's golden
        vectors at 1e-4), with 2 GFLOP instead of 70 and no N x 1024 tensor anywhere."""
        bsz, tgt_len, _ = v.size()
        H, hd = self.num_heads, self.head_dim
        if self.reassociate:
            src_len = l.size(1)
            fused = self.fused_softmax and bi_softmax_supported(v, H, src_len, self.training and self.dropout > 0)
            composed = self._composed_text_side(l) if fused else None
            if composed is not None:   # the text side's double projections as one constant matrix each (see there)
                AC, ac0, Z, z0, O, o0 = composed[:6]
                l2 = l.reshape(bsz * src_len, -1)
                ac = torch.addmm(ac0, l2, AC).view(bsz, src_len, -1)
                a = ac[..., :H * self.v_dim].reshape(bsz, src_len, H, self.v_dim).permute(0, 3, 2, 1)   # [B, v_dim, H, T]
                c = ac[..., H * self.v_dim:].permute(0, 2, 1)                                            # [B, H, T]
                value_l4 = k4 = None
            else:
                k4 = self.l_proj(l).view(bsz, src_len, H, hd)
                value_l4 = self.values_l_proj(l).view(bsz, src_len, H, hd)
                wq = self.v_proj.weight.view(H, hd, -1)
                a = torch.einsum("hed,bthe->bdht", wq, k4) * self.scale             # [B, v_dim, H, T]
                c = torch.einsum("he,bthe->bht", self.v_proj.bias.view(H, hd), k4) * self.scale
            if fused:
                # everything between the score GEMM and the two output GEMMs in one HIP op
                # (csrc/bisoftmax.hip), tensors staying in the GEMMs' [B, N, H*T] layout
                xm = wide_matmul(v, a.reshape(bsz, -1, H * src_len))
                pv, e, colsum = bi_softmax(xm, c.reshape(bsz, H * src_len), attention_mask_l, attention_mask_v,
                                           H, src_len, self.stable_softmax_2d, self.clamp_min_for_underflow,
                                           self.clamp_max_for_overflow)
                u = (tall_reduce_nt(e, v) / colsum[..., None]).view(bsz, H, src_len, -1)   # P_l v
                if composed is not None:
                    out_l = torch.addmm(o0, u.permute(0, 2, 1, 3).reshape(bsz * src_len, -1), O).view(bsz, src_len, -1)
                    z = torch.addmm(z0, l2, Z).view(bsz, src_len, H, -1).permute(0, 2, 1, 3)          # [B, H, T, v_dim]
                else:
                    out_l = torch.einsum("bhtd,hed->bthe", u, self.values_v_proj.weight.view(H, hd, -1))
                    out_l = out_l + self.values_v_proj.bias.view(H, hd)     # rows of P_l sum to one
                    out_l = self.out_l_proj(out_l.reshape(bsz, src_len, self.embed_dim))
                    z = torch.einsum("bthe,dhe->bhtd", value_l4, self.out_v_proj.weight.view(-1, H, hd))
                z = z.reshape(bsz, H * src_len, -1)
                if residual_v is not None:
                    scale = residual_v()
                    if wide_matmul_residual_supported(pv, z, self.out_v_proj.bias, v, scale):
                        return wide_matmul_residual(pv, z, self.out_v_proj.bias, v, scale), out_l, True
                    return torch.addcmul(v, wide_matmul(pv, z, self.out_v_proj.bias), scale), out_l, True
                out_v = wide_matmul(pv, z, self.out_v_proj.bias)
                return out_v, out_l
            attn = wide_matmul(v, a.reshape(bsz, -1, H * src_len)).view(bsz, tgt_len, H, src_len) + c[:, None]
            attn = attn.permute(0, 2, 1, 3).reshape(bsz * H, tgt_len, src_len)  # [bs*heads, n_img, n_text]
        else:
            q = self._heads(self.v_proj(v) * self.scale, bsz)
            k = self._heads(self.l_proj(l), bsz)
            value_v = self._heads(self.values_v_proj(v), bsz)
            value_l = self._heads(self.values_l_proj(l), bsz)
            src_len = k.size(1)
            attn = torch.bmm(q, k.transpose(1, 2))  # [bs*heads, n_img, n_text]
        if self.stable_softmax_2d:
            attn = _SubtractGlobalMax.apply(attn)
        attn = self._clamp(attn)

        attn_T = attn.transpose(1, 2)
        # reference: attn_T - torch.max(attn_T, dim=-1, keepdim=True)[0]  (fuse_modules.py:180)
        attn_l = attn_T - _max_over_tokens(attn).transpose(1, 2)
        attn_l = self._clamp(attn_l)
        if attention_mask_v is not None:
            mv = attention_mask_v[:, None, None, :].repeat(1, self.num_heads, 1, 1).flatten(0, 1)
            attn_l = attn_l.masked_fill(mv, float("-inf"))
        attn_l = attn_l.softmax(dim=-1)

        if attention_mask_l is not None:
            ml = attention_mask_l[:, None, None, :].repeat(1, self.num_heads, 1, 1).flatten(0, 1)
            attn = attn.masked_fill(ml, float("-inf"))
        attn_v = attn.softmax(dim=-1)

        probs_v = F.dropout(attn_v, p=self.dropout, training=self.training)
        probs_l = F.dropout(attn_l, p=self.dropout, training=self.training)
        if self.reassociate:
            wv = self.values_v_proj.weight.view(H, hd, -1)
            u = tall_reduce(probs_l.reshape(bsz, H * src_len, tgt_len), v).view(bsz, H, src_len, -1)
            out_l = torch.einsum("bhtd,hed->bthe", u, wv)
            out_l = out_l + (probs_l.sum(-1).view(bsz, H, src_len).transpose(1, 2)[..., None]
                             * self.values_v_proj.bias.view(H, hd))
            out_l = self.out_l_proj(out_l.reshape(bsz, src_len, self.embed_dim))
            wo = self.out_v_proj.weight.view(-1, H, hd)
            z = torch.einsum("bthe,dhe->bhtd", value_l4, wo).reshape(bsz, H * src_len, -1)
            pv = probs_v.view(bsz, H, tgt_len, src_len).permute(0, 2, 1, 3).reshape(bsz, tgt_len, H * src_len)
            out_v = wide_matmul(pv, z, self.out_v_proj.bias)
            return out_v, out_l
        out_v = torch.bmm(probs_v, value_l)
        out_l = torch.bmm(probs_l, value_v)
        out_v = out_v.view(bsz, self.num_heads, tgt_len, self.head_dim).transpose(1, 2).reshape(
            bsz, tgt_len, self.embed_dim)
        out_l = out_l.view(bsz, self.num_heads, src_len, self.head_dim).transpose(1, 2).reshape(
            bsz, src_len, self.embed_dim)
        return self.out_v_proj(out_v), self.out_l_proj(out_l)


class BiAttentionBlock(nn.Module):
    """Pre-LN bi-directional fusion with layer-scale gammas (reference fuse_modules.py:252-305)."""

    def __init__(self, v_dim, l_dim, embed_dim, num_heads, dropout=0.1, drop_path=0.0,
                 init_values=1e-4, cfg=None):
        super().__init__()
        self.layer_norm_v = LayerNorm(v_dim)
        self.layer_norm_l = LayerNorm(l_dim)
        self.attn = BiMultiHeadAttention(v_dim=v_dim, l_dim=l_dim, embed_dim=embed_dim,
                                         num_heads=num_heads, dropout=dropout)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.gamma_v = nn.Parameter(init_values * torch.ones((v_dim)), requires_grad=True)
        self.gamma_l = nn.Parameter(init_values * torch.ones((l_dim)), requires_grad=True)

    def forward(self, v, l, attention_mask_v=None, attention_mask_l=None):
        v = self.layer_norm_v(v)
        if self.native_text_side and self.fused_residual and self.residual_in_gemm and not self.gamma_v.requires_grad:
            got = self._forward_native_text(v, l, attention_mask_v, attention_mask_l)
            if got is not None:
                return got
        l = self.layer_norm_l(l)
        if not self.fused_residual:
            delta_v, delta_l = self.attn(v, l, attention_mask_v=attention_mask_v, attention_mask_l=attention_mask_l)
            return v + self.drop_path(self.gamma_v * delta_v), l + self.drop_path(self.gamma_l * delta_l)
        # the image-side residual rides in the attention's last GEMM when it can (the scale is drawn when that GEMM is
        # reached: after everything the attention itself draws, as in the reference's order)
        drawn = []     # the two stochastic-depth factors of this block come from ONE draw ([2, B, 1, 1]: image side, text side)

        def scale_of(gamma, x, side):
            dp = self.drop_path
            if not (isinstance(dp, DropPath) and dp.drop_prob > 0.0 and self.training):
                return gamma
            if not drawn:
                keep = 1 - dp.drop_prob
                drawn.append(x.new_empty((2, x.shape[0]) + (1,) * (x.ndim - 1)).bernoulli_(keep).div_(keep))
            return gamma * drawn[0][side]

        in_gemm = (lambda: scale_of(self.gamma_v, v, 0)) if self.residual_in_gemm and not self.gamma_v.requires_grad else None
        got = self.attn(v, l, attention_mask_v=attention_mask_v, attention_mask_l=attention_mask_l, residual_v=in_gemm)
        if len(got) == 3:
            return got[0], torch.addcmul(l, got[1], scale_of(self.gamma_l, l, 1))
        return torch.addcmul(v, got[0], scale_of(self.gamma_v, v, 0)), torch.addcmul(l, got[1], scale_of(self.gamma_l, l, 1))

    fused_residual = True     # class-level switches for A/B runs
    residual_in_gemm = True
    native_text_side = True   # frozen composed projections: the text side as two native nodes (text_side.py)
    fused_image_side = True   # ... and the image side as one (dense._FusionImageSide; H T <= 128)

    def _forward_native_text(self, v, l, attention_mask_v, attention_mask_l):
        """The fused, re-bracketed block with its text side on csrc/textside.hip: LayerNorm of the text, the composed
        projections in the layouts the image-side GEMMs read, and the text output with its residual -- 6 launches instead of
        ~41 (same arithmetic up to fp32 summation order; ``v`` is already normalised).  None when it does not apply."""
        from . import text_side
        att = self.attn
        bsz, _, _ = v.shape
        T, H = l.size(1), att.num_heads
        if not (att.reassociate and l.is_cuda and att.fused_softmax
                and bi_softmax_supported(v, H, T, att.training and att.dropout > 0)):
            return None
        composed = att._composed_text_side(l)
        if not text_side.supported(l, self.layer_norm_l, self.gamma_l, composed):
            return None
        O, o0, W1, b1, W1T, OT = composed[4:10]
        dp = self.drop_path
        keep = None
        if isinstance(dp, DropPath) and dp.drop_prob > 0.0 and self.training:
            kp = 1 - dp.drop_prob
            keep = v.new_empty((2, bsz)).bernoulli_(kp).div_(kp)      # one draw: image side, text side
        l_ln, a, c, z = text_side.text_prep(l, self.layer_norm_l, W1, b1, W1T, H, att.v_dim)
        scale = self.gamma_v if keep is None else self.gamma_v * keep[0].view(bsz, 1, 1)
        if self.fused_image_side and fusion_image_side_supported(v, a, z, att.out_v_proj.bias, scale, H, T, False):
            # the four image-side ops below as one autograd node (the gradient of v in one pass, dense._FusionImageSide)
            out_v, t, colsum = fusion_image_side(v, a, c, z, att.out_v_proj.bias, scale, attention_mask_l, attention_mask_v, H, T,
                                                 att.stable_softmax_2d, att.clamp_min_for_underflow, att.clamp_max_for_overflow)
            return out_v, text_side.text_out(t, colsum, l_ln, O, OT, o0, self.gamma_l, None if keep is None else keep[1], H)
        xm = wide_matmul(v, a)
        pv, e, colsum = bi_softmax(xm, c, attention_mask_l, attention_mask_v, H, T, att.stable_softmax_2d,
                                   att.clamp_min_for_underflow, att.clamp_max_for_overflow)
        out_l = text_side.text_out(tall_reduce_nt(e, v), colsum, l_ln, O, OT, o0, self.gamma_l, None if keep is None else keep[1], H)
        if wide_matmul_residual_supported(pv, z, att.out_v_proj.bias, v, scale):
            return wide_matmul_residual(pv, z, att.out_v_proj.bias, v, scale), out_l
        return torch.addcmul(v, wide_matmul(pv, z, att.out_v_proj.bias), scale), out_l

    # (x + drop_path(gamma * delta) is one addcmul: the layer scale and the per-sample stochastic-depth factor are folded into a
    #  [B, 1, C] scale first -- the reference's three elementwise passes over the [B, S, 256] image tokens, scale, mask, add, move
    #  2.7x the bytes.  Round 5: the two factors of a block come from one bernoulli call -- same distribution, two launches fewer.)


class TransformerEncoderLayer(nn.Module):
    """Text-enhancer layer (reference transformer_vanilla.py:72-123): post-LN self-attention
    over the tokens with the block-diagonal sub-sentence mask, then FFN."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu",
                 normalize_before=False):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.dropout = nn.Dropout(dropout)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = LayerNorm(d_model)
        self.norm2 = LayerNorm(d_model)
        self.dropout1 = nn.Dropout(dropout)
        self.dropout2 = nn.Dropout(dropout)
        self.activation = _get_activation_fn(activation)
        self.normalize_before = normalize_before
        self.nhead = nhead

    @staticmethod
    def with_pos_embed(tensor, pos: Optional[Tensor]):
        return tensor if pos is None else tensor + pos

    def forward(self, src, src_mask: Optional[Tensor] = None,
                src_key_padding_mask: Optional[Tensor] = None, pos: Optional[Tensor] = None):
        # [bs, T, T] -> [nhead*bs, T, T] by tiling, exactly as the reference does (:109-112).
        # nn.MultiheadAttention indexes that dimension as b*nhead + h, so with bs > 1 the
        # masks of different images are interleaved over heads; kept for output parity.
        if src_mask.dim() == 3 and src_mask.shape[0] == src.shape[1]:
            src_mask = src_mask.repeat(self.nhead, 1, 1)
        q = k = self.with_pos_embed(src, pos)
        src2 = _mha(self.self_attn, q, k, src, attn_mask=src_mask)
        src = self.norm1(src + self.dropout1(src2))
        src2 = self.linear2(self.dropout(self.activation(self.linear1(src))))
        return self.norm2(src + self.dropout2(src2))

    # (A row-GEMM form of this layer -- three autograd nodes round the attention core, 43 -> 31 launches, 267 -> 206 us by kernel
    #  time -- was built in round 4 and REMOVED in round 5: the layer runs beside the deformable image layer, off the critical
    #  path, and the step did not change with 32 text tokens (37.4 ms either way) nor with 194 (67.1-67.4 against 66.7-67.4 ms).)


class DeformableTransformerEncoderLayer(nn.Module):
    """MSDA self-attention over all pixels of all levels (Q = S) + FFN, post-LN
    (reference transformer_for_adapter.py:809-907)."""

    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4,
                 n_heads=8, n_points=4, use_adapter=False, **_unused):
        super().__init__()
        if use_adapter:
            raise NotImplementedError("use_adapter (Adapter/MoE ablation) is outside the ZiRa path")
        self.self_attn = MSDeformAttn(embed_dim=d_model, num_levels=n_levels, num_heads=n_heads,
                                      num_points=n_points, batch_first=True)
        self.dropout1 = nn.Dropout(dropout)
        self.norm1 = LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = _get_activation_fn(activation, d_model=d_ffn)
        self.dropout2 = nn.Dropout(dropout)
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout3 = nn.Dropout(dropout)
        self.norm2 = LayerNorm(d_model)
        self.use_adapter = False

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def refresh_fused_projection(self, *unused):
        """The bf16 planes of the FFN weights (``Switches.gemm_arith`` = "bf16x3") follow the parameters in place."""
        for sp in (getattr(self, "_ffn_split_weights", None), getattr(self, "_ffn_fused_weights", None)):
            if sp is not None:
                sp.refresh(self.linear1, self.linear2)

    fuse_bias_relu = True   # bias + ReLU in the GEMM epilogue: -117 us per layer (scripts/enclayer_profile.py)
    native_attention = True   # frozen fp32 GPU calls without padding: the attention sublayer as one autograd node (encoder_layer.py)

    def forward_ffn(self, src):
        if (self.fuse_bias_relu and self.activation is F.relu and src.is_cuda and src.dtype == torch.float32
                and (self.dropout2.p == 0.0 or not self.training) and not torch.is_autocast_enabled()):
            x2 = src.reshape(-1, src.shape[-1])
            if self.dropout3.p == 0.0 and _frozen_ffn_norm_ok(x2, self.linear1, self.linear2, self.norm2):
                return _FrozenFFNNorm.apply(x2, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias,
                                            self.norm2.weight, self.norm2.bias, self.norm2.eps,
                                            _ffn_split(self, x2)).view_as(src), src.new_zeros(1)
            if _frozen_ffn_ok(x2, self.linear1, self.linear2):
                src2 = _FrozenFFN.apply(x2, self.linear1.weight, self.linear1.bias, self.linear2.weight,
                                        self.linear2.bias, _ffn_split(self, x2)).view(*src.shape[:-1], -1)
            else:
                h = _LinearReLU.apply(x2, self.linear1.weight, self.linear1.bias)
                src2 = self.linear2(h.view(*src.shape[:-1], -1))
        else:
            src2 = self.linear2(self.dropout2(self.activation(self.linear1(src))))
        return self.norm2.add_norm(src, self.dropout3(src2)), src.new_zeros(1)   # (residual add inside the LN kernel)

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index,
                key_padding_mask=None):
        if self.native_attention and src.is_cuda:
            from . import encoder_layer as native
            if native.applies(self, src, pos, reference_points, spatial_shapes, key_padding_mask):
                # projections, sampling plan, MSDA, output projection, residual + LayerNorm: one autograd node whose backward
                # lets the three gradients of src meet inside its GEMMs (encoder_layer.py)
                return self.forward_ffn(native.attention_sublayer(self, src, pos, reference_points, spatial_shapes,
                                                                  level_start_index))
        # (query = src + pos, value = src: handed over as ONE tensor + pos so that the module can treat them as one node)
        src2 = self.self_attn(query=src, query_pos=pos, reference_points=reference_points,
                              value=src, spatial_shapes=spatial_shapes,
                              level_start_index=level_start_index, key_padding_mask=key_padding_mask)
        src = self.norm1.add_norm(src, self.dropout1(src2))
        return self.forward_ffn(src)


class DeformableTransformerDecoderLayer(nn.Module):
    """self-attn -> text cross-attn -> MSDA cross-attn -> FFN, post-LN
    (reference transformer_for_adapter.py:910-1073)."""

    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4,
                 n_heads=8, n_points=4, use_text_feat_guide=False, use_text_cross_attention=False,
                 use_adapter=False, **_unused):
        super().__init__()
        if use_adapter:
            raise NotImplementedError("use_adapter (Adapter/MoE ablation) is outside the ZiRa path")
        assert not use_text_feat_guide
        ident_or_drop = lambda: nn.Dropout(dropout) if dropout > 0 else nn.Identity()
        self.cross_attn = MSDeformAttn(embed_dim=d_model, num_levels=n_levels, num_heads=n_heads,
                                       num_points=n_points, batch_first=True)
        self.dropout1 = ident_or_drop()
        self.norm1 = LayerNorm(d_model)
        if use_text_cross_attention:
            self.ca_text = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
            self.catext_dropout = ident_or_drop()
            self.catext_norm = LayerNorm(d_model)
        self.self_attn = nn.MultiheadAttention(d_model, n_heads, dropout=dropout)
        self.dropout2 = ident_or_drop()
        self.norm2 = LayerNorm(d_model)
        self.linear1 = nn.Linear(d_model, d_ffn)
        self.activation = _get_activation_fn(activation, d_model=d_ffn, batch_dim=1)
        self.dropout3 = ident_or_drop()
        self.linear2 = nn.Linear(d_ffn, d_model)
        self.dropout4 = ident_or_drop()
        self.norm3 = LayerNorm(d_model)
        self.key_aware_proj = None
        self.use_text_feat_guide = use_text_feat_guide
        self.use_text_cross_attention = use_text_cross_attention
        self.use_adapter = False

    def rm_self_attn_modules(self):
        self.self_attn = None
        self.dropout2 = None
        self.norm2 = None

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    fuse_bias_relu = True   # bias + ReLU in the GEMM epilogue, as in the encoder layer
    native_layer = True     # frozen fp32 GPU calls: the layer as one autograd node (decoder_layer.py); False: the module composition

    def refresh_fused_projection(self, *unused):
        """Weights the native path keeps transposed follow the parameters in place (GraphedTransformer calls this when a
        parameter changed: a replayed graph re-runs no Python)."""
        w = getattr(self, "_native_weights", None)
        if w is not None:
            w.refresh(self)

    def forward_ffn(self, tgt):
        with torch.amp.autocast("cuda", enabled=False):  # reference :1004 keeps the FFN in fp32
            if (self.fuse_bias_relu and self.activation is F.relu and tgt.is_cuda and tgt.dtype == torch.float32
                    and (getattr(self.dropout3, "p", 0.0) == 0.0 or not self.training)):
                x2 = tgt.reshape(-1, tgt.shape[-1])
                if _frozen_ffn_ok(x2, self.linear1, self.linear2):
                    tgt2 = _FrozenFFN.apply(x2, self.linear1.weight, self.linear1.bias, self.linear2.weight,
                                            self.linear2.bias).view(*tgt.shape[:-1], -1)
                else:
                    h = _LinearReLU.apply(x2, self.linear1.weight, self.linear1.bias)
                    tgt2 = self.linear2(h.view(*tgt.shape[:-1], -1))
            else:
                tgt2 = self.linear2(self.dropout3(self.activation(self.linear1(tgt))))
        return self.norm3(tgt + self.dropout4(tgt2)), tgt.new_zeros(1)

    def forward(self, tgt, tgt_query_pos=None, tgt_query_sine_embed=None, tgt_key_padding_mask=None,
                tgt_reference_points=None, memory_text=None, text_attention_mask=None, memory=None,
                memory_key_padding_mask=None, memory_level_start_index=None,
                memory_spatial_shapes=None, memory_pos=None, self_attn_mask=None,
                cross_attn_mask=None, memory_text_lb=None, memory_value=None, tgt_reference_points_bf=None):
        assert cross_attn_mask is None
        if self.native_layer and memory_value is not None:
            # every weight frozen (a ZiRa task), fp32 on the GPU: the whole layer is one autograd node whose launches carry the
            # adds, LayerNorms and their gradients in the GEMMs' prologues / epilogues (decoder_layer.py)
            from . import decoder_layer as native
            if native.applies(self, tgt, tgt_query_pos, tgt_reference_points, memory_text, memory_value, self_attn_mask,
                              tgt_key_padding_mask):
                text_lb = memory_text_lb if memory_text_lb is not None else memory_text.transpose(0, 1)
                kpm = text_attention_mask
                if kpm is not None and kpm.dtype == torch.bool:
                    kpm = _additive_mask(kpm, tgt.dtype)
                out = native.decoder_layer_forward(self, tgt, tgt_query_pos, tgt_reference_points, text_lb, kpm, memory_value,
                                                   memory_spatial_shapes, memory_level_start_index,
                                                   ref_bf=tgt_reference_points_bf)
                return out, None
        if self.self_attn is not None:
            q = k = self.with_pos_embed(tgt, tgt_query_pos)
            tgt2 = _mha(self.self_attn, q, k, tgt, attn_mask=self_attn_mask)
            tgt = self.norm2(tgt + self.dropout2(tgt2))
        if self.use_text_cross_attention:
            # (the decoder passes the [tokens, batch, C] copy it made once for all its layers)
            text_lb = memory_text_lb if memory_text_lb is not None else memory_text.transpose(0, 1)
            tgt2 = _mha(self.ca_text, self.with_pos_embed(tgt, tgt_query_pos), text_lb, text_lb,
                        key_padding_mask=text_attention_mask)
            tgt = self.catext_norm(tgt + self.catext_dropout(tgt2))
        tgt2 = self.cross_attn(
            query=self.with_pos_embed(tgt, tgt_query_pos).transpose(0, 1),
            reference_points=tgt_reference_points.transpose(0, 1).contiguous(),
            value=memory.transpose(0, 1), spatial_shapes=memory_spatial_shapes,
            level_start_index=memory_level_start_index, key_padding_mask=memory_key_padding_mask,
            value_projected=memory_value,   # (this layer's value_proj(memory), made for all layers at once by the decoder)
        ).transpose(0, 1)
        tgt = self.norm1(tgt + self.dropout1(tgt2))
        return self.forward_ffn(tgt)


class TransformerEncoder(nn.Module):
    """Per layer: BiAttention fusion -> text enhancer -> deformable image layer
    (reference transformer_for_adapter.py:423-662)."""

    def __init__(self, encoder_layer, num_layers, d_model=256, num_queries=300,
                 enc_layer_share=False, text_enhance_layer=None, feature_fusion_layer=None,
                 use_checkpoint=False, use_transformer_ckpt=False):
        super().__init__()
        self.layers, self.text_layers, self.fusion_layers = [], [], []
        if num_layers > 0:
            self.layers = _get_clones(encoder_layer, num_layers, layer_share=enc_layer_share)
            if text_enhance_layer is not None:
                self.text_layers = _get_clones(text_enhance_layer, num_layers, layer_share=enc_layer_share)
            if feature_fusion_layer is not None:
                self.fusion_layers = _get_clones(feature_fusion_layer, num_layers, layer_share=enc_layer_share)
        self.query_scale = None
        self.num_queries = num_queries
        self.num_layers = num_layers
        self.d_model = d_model
        self.use_checkpoint = use_checkpoint
        self.use_transformer_ckpt = use_transformer_ckpt

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device):
        """Pixel centres of every level, normalised by the valid extent and replicated over the
        levels: [bs, S, L, 2] (reference :482-497).  ``spatial_shapes``: list of (H, W) or tensor."""
        shapes = spatial_shapes.tolist() if torch.is_tensor(spatial_shapes) else spatial_shapes
        if (Switches.native_geometry and valid_ratios.is_cuda and valid_ratios.dtype == torch.float32
                and not valid_ratios.requires_grad and 0 < len(shapes) <= 30):
            from . import geometry   # one launch instead of 38 (csrc/refpoints.hip), bit-identical
            return geometry.encoder_reference_points(valid_ratios.contiguous(), shapes)
        refs = []
        for lvl, (H_, W_) in enumerate(shapes):
            H_, W_ = int(H_), int(W_)
            ref_y, ref_x = torch.meshgrid(
                torch.linspace(0.5, H_ - 0.5, H_, dtype=torch.float32, device=device),
                torch.linspace(0.5, W_ - 0.5, W_, dtype=torch.float32, device=device), indexing="ij")
            ref_y = ref_y.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H_)
            ref_x = ref_x.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W_)
            refs.append(torch.stack((ref_x, ref_y), -1))
        reference_points = torch.cat(refs, 1)
        return reference_points[:, :, None] * valid_ratios[:, None]

    def forward(self, src: Tensor, pos: Tensor, spatial_shapes: Tensor, level_start_index: Tensor,
                valid_ratios: Tensor, key_padding_mask: Tensor, memory_text: Tensor = None,
                text_attention_mask: Tensor = None, pos_text: Tensor = None,
                text_self_attention_masks: Tensor = None, position_ids: Tensor = None,
                spatial_shapes_list=None):
        reference_points, pos_text = self.prepare(
            spatial_shapes_list if spatial_shapes_list is not None else spatial_shapes, valid_ratios,
            memory_text, pos_text, position_ids, src.device)
        output = src
        for layer_id in range(len(self.layers)):
            output, memory_text = self.forward_layer(
                layer_id, output, memory_text, pos, reference_points, spatial_shapes, level_start_index,
                key_padding_mask, text_attention_mask, pos_text, text_self_attention_masks)
        return output, memory_text, src.new_zeros(1)

    def prepare(self, shapes, valid_ratios, memory_text, pos_text, position_ids, device):
        """Per-call constants of the layer loop: pixel reference points and text position codes."""
        reference_points = None
        if self.num_layers > 0:
            reference_points = self.get_reference_points(shapes, valid_ratios, device=device)
        if self.text_layers:
            bs, n_text, _ = memory_text.shape
            if pos_text is None and position_ids is None:
                pos_text = (torch.arange(n_text, device=memory_text.device).float()
                            .unsqueeze(0).unsqueeze(-1).repeat(bs, 1, 1))
                pos_text = get_sine_pos_embed(pos_text, num_pos_feats=256, exchange_xy=False)
            if position_ids is not None:
                pos_text = get_sine_pos_embed(position_ids[..., None], num_pos_feats=256, exchange_xy=False)
        return reference_points, pos_text

    def forward_layer(self, layer_id, output, memory_text, pos, reference_points, spatial_shapes,
                      level_start_index, key_padding_mask, text_attention_mask, pos_text,
                      text_self_attention_masks, fuse=True):
        """One encoder layer: fusion -> text enhancer -> deformable image layer (reference :563-662).
        ``fuse=False`` skips the fusion block (the caller has already applied it)."""
        if self.fusion_layers and fuse:
            output, memory_text = self.fusion_layers[layer_id](
                v=output, l=memory_text, attention_mask_v=key_padding_mask,
                attention_mask_l=text_attention_mask)
        def text_layer(mt):
            return self.text_layers[layer_id](
                src=mt.transpose(0, 1),
                src_mask=~text_self_attention_masks,  # True = do not attend
                src_key_padding_mask=text_attention_mask,
                pos=(pos_text.transpose(0, 1) if pos_text is not None else None),
            ).transpose(0, 1)

        side = None
        if self.text_layers:
            if self.overlap_text_layer and output.is_cuda:
                # The text enhancer (~60 launch-bound kernels on <= 256 tokens) and the deformable image layer are
                # independent inside an encoder layer: the text side goes to a second stream (captured as a parallel
                # branch when the layer is replayed from a hipGraph; autograd runs its backward on that stream too).
                cur = torch.cuda.current_stream(output.device)
                side = self._text_stream(output.device)
                side.wait_stream(cur)
                # The text tokens were allocated on `cur` and are read on `side`: without gradients (eval, or a no-grad pass)
                # nothing keeps them alive once `memory_text` is rebound below, and the allocator of `cur` would hand their
                # block to the image layer running beside -- found in round 5 as an intermittently wrong two-stage selection
                # of the frozen full-size transformer (scripts/repro_frozen_nograd.py: 5 of 12 fresh models, 0 of 12 since).
                memory_text.record_stream(side)
                with torch.cuda.stream(side):
                    memory_text = text_layer(memory_text)
            else:
                memory_text = text_layer(memory_text)
        output, _ = self.layers[layer_id](src=output, pos=pos, reference_points=reference_points,
                                          spatial_shapes=spatial_shapes,
                                          level_start_index=level_start_index,
                                          key_padding_mask=key_padding_mask)
        if side is not None:
            cur.wait_stream(side)
            memory_text.record_stream(cur)
        return output, memory_text

    overlap_text_layer = True
    _side_streams = {}

    @classmethod
    def _text_stream(cls, device):
        s = cls._side_streams.get(device)
        if s is None:
            s = cls._side_streams[device] = torch.cuda.Stream(device=device)
        return s


class TransformerDecoder(nn.Module):
    """Six layers with iterative box refinement (reference transformer_for_adapter.py:665-806)."""

    batch_value_projections = True   # class-level switch (tests compare both ways)
    native_glue = True               # frozen fp32 GPU calls: box refinement + intermediate LayerNorm as one node (decoder_layer.py)

    def __init__(self, decoder_layer, num_layers, norm=None, return_intermediate=False, d_model=256,
                 query_dim=4, num_feature_levels=1):
        super().__init__()
        self.layers = _get_clones(decoder_layer, num_layers) if num_layers > 0 else []
        self.num_layers = num_layers
        self.norm = norm
        self.return_intermediate = return_intermediate
        assert return_intermediate, "support return_intermediate only"
        assert query_dim in [2, 4]
        self.query_dim = query_dim
        self.num_feature_levels = num_feature_levels
        self.ref_point_head = MLP(query_dim // 2 * d_model, d_model, d_model, 2)
        self.query_pos_sine_scale = None
        self.query_scale = None
        self.bbox_embed = None
        self.class_embed = None
        self.d_model = d_model
        self.ref_anchor_head = None

    def forward(self, tgt, memory, tgt_mask=None, memory_mask=None, tgt_key_padding_mask=None,
                memory_key_padding_mask=None, pos=None, refpoints_unsigmoid=None,
                level_start_index=None, spatial_shapes=None, valid_ratios=None, memory_text=None,
                text_attention_mask=None):
        output = tgt
        intermediate = []
        reference_points = refpoints_unsigmoid.sigmoid()
        ref_points = [reference_points]
        adapter_loss = tgt.new_zeros(1)
        # the same for every layer: the ratios of a 4-d reference point and the [tokens, batch, C] text memory
        ratios4 = torch.cat([valid_ratios, valid_ratios], -1)[None, :] if reference_points.shape[-1] == 4 else None
        memory_text_lb = memory_text.transpose(0, 1).contiguous() if memory_text is not None else None
        # ... and the value projections of all layers' deformable cross-attention: one autograd node whose input gradient
        # is accumulated by the GEMMs themselves (None: the layers project for themselves)
        memory_values = multi_value_projections([layer.cross_attn for layer in self.layers], memory.transpose(0, 1),
                                                memory_key_padding_mask) if self.batch_value_projections else None
        native = None
        if self.native_glue and output.is_cuda:
            from . import decoder_layer as native
        for layer_id, layer in enumerate(self.layers):
            ref_bf = None
            if native is not None and native.prep_applies(self, reference_points, valid_ratios):
                # boxes per level (both layouts), their sine embedding and the position MLP: three launches, no autograd
                reference_points_input, ref_bf, query_sine_embed, query_pos = native.prep_queries(self, reference_points, valid_ratios)
            else:
                if reference_points.shape[-1] == 4:
                    reference_points_input = reference_points[:, :, None] * ratios4
                else:
                    reference_points_input = reference_points[:, :, None] * valid_ratios[None, :]
                query_sine_embed = gen_sineembed_for_position(reference_points_input[:, :, 0, :])
                raw_query_pos = self.ref_point_head(query_sine_embed)
                query_pos = raw_query_pos if self.query_scale is None else self.query_scale(output) * raw_query_pos

            output, adapter_loss_ = layer(
                tgt=output, tgt_query_pos=query_pos, tgt_query_sine_embed=query_sine_embed,
                tgt_key_padding_mask=tgt_key_padding_mask,
                tgt_reference_points=reference_points_input, memory_text=memory_text,
                text_attention_mask=text_attention_mask, memory=memory,
                memory_key_padding_mask=memory_key_padding_mask,
                memory_level_start_index=level_start_index, memory_spatial_shapes=spatial_shapes,
                memory_pos=pos, self_attn_mask=tgt_mask, cross_attn_mask=memory_mask,
                memory_text_lb=memory_text_lb,
                memory_value=None if memory_values is None else memory_values[layer_id],
                tgt_reference_points_bf=ref_bf)
            if adapter_loss_ is not None:   # (the native layer path has no adapter term and returns None for it)
                adapter_loss = adapter_loss + adapter_loss_

            if self.native_glue and self.bbox_embed is not None and output.is_cuda:
                from . import decoder_layer as native
                if native.refine_applies(self, layer_id, output, reference_points):
                    # box MLP + inverse-sigmoid + sigmoid and the LayerNorm of the intermediate output: one autograd node
                    new_reference_points, normed = native.refine_and_norm(self, layer_id, output, reference_points)
                    reference_points = new_reference_points.detach()
                    ref_points.append(new_reference_points)
                    intermediate.append(normed)
                    continue
            if self.bbox_embed is not None:  # iterative refinement, detached between layers
                delta_unsig = self.bbox_embed[layer_id](output)
                new_reference_points = (delta_unsig + inverse_sigmoid(reference_points)).sigmoid()
                reference_points = new_reference_points.detach()
                ref_points.append(new_reference_points)
            intermediate.append(self.norm(output))
        return [[x.transpose(0, 1) for x in intermediate],
                [r.transpose(0, 1) for r in ref_points], adapter_loss]


class Transformer(nn.Module):
    """Encoder -> two-stage query selection (top-900 by max token logit) -> decoder
    (reference transformer_for_adapter.py:41-415)."""

    def __init__(self, d_model=256, nhead=8, num_queries=300, num_encoder_layers=6,
                 num_unicoder_layers=0, num_decoder_layers=6, dim_feedforward=2048, dropout=0.0,
                 activation="relu", normalize_before=False, return_intermediate_dec=False,
                 query_dim=4, num_patterns=0, num_feature_levels=1, enc_n_points=4, dec_n_points=4,
                 learnable_tgt_init=False, two_stage_type="no", embed_init_tgt=False,
                 use_text_enhancer=False, use_fusion_layer=False, use_checkpoint=False,
                 use_transformer_ckpt=False, use_text_cross_attention=False, text_dropout=0.1,
                 fusion_dropout=0.1, fusion_droppath=0.0, use_adapter=False, **_unused):
        super().__init__()
        assert query_dim == 4
        assert not normalize_before
        assert learnable_tgt_init, "why not learnable_tgt_init"
        assert two_stage_type in ["no", "standard"]
        self.num_feature_levels = num_feature_levels
        self.num_encoder_layers = num_encoder_layers
        self.num_unicoder_layers = num_unicoder_layers
        self.num_decoder_layers = num_decoder_layers
        self.num_queries = num_queries

        encoder_layer = DeformableTransformerEncoderLayer(
            d_model, dim_feedforward, dropout, activation, num_feature_levels, nhead, enc_n_points,
            use_adapter=use_adapter)
        text_enhance_layer = TransformerEncoderLayer(
            d_model=d_model, nhead=nhead // 2, dim_feedforward=dim_feedforward // 2,
            dropout=text_dropout) if use_text_enhancer else None
        feature_fusion_layer = BiAttentionBlock(
            v_dim=d_model, l_dim=d_model, embed_dim=dim_feedforward // 2, num_heads=nhead // 2,
            dropout=fusion_dropout, drop_path=fusion_droppath) if use_fusion_layer else None
        self.encoder = TransformerEncoder(
            encoder_layer, num_encoder_layers, d_model=d_model, num_queries=num_queries,
            text_enhance_layer=text_enhance_layer, feature_fusion_layer=feature_fusion_layer,
            use_checkpoint=use_checkpoint, use_transformer_ckpt=use_transformer_ckpt)

        decoder_layer = DeformableTransformerDecoderLayer(
            d_model, dim_feedforward, dropout, activation, num_feature_levels, nhead, dec_n_points,
            use_text_cross_attention=use_text_cross_attention, use_adapter=use_adapter)
        self.decoder = TransformerDecoder(
            decoder_layer, num_decoder_layers, LayerNorm(d_model),
            return_intermediate=return_intermediate_dec, d_model=d_model, query_dim=query_dim,
            num_feature_levels=num_feature_levels)

        self.d_model = d_model
        self.nhead = nhead
        self.dec_layers = num_decoder_layers
        self.num_patterns = num_patterns if isinstance(num_patterns, int) else 0
        if num_feature_levels > 1:
            self.level_embed = (nn.Parameter(torch.Tensor(num_feature_levels, d_model))
                                if num_encoder_layers > 0 else None)
        self.learnable_tgt_init = learnable_tgt_init
        self.embed_init_tgt = embed_init_tgt
        if (two_stage_type != "no" and embed_init_tgt) or two_stage_type == "no":
            self.tgt_embed = nn.Embedding(self.num_queries, d_model)
            nn.init.normal_(self.tgt_embed.weight.data)
        else:
            self.tgt_embed = None
        self.two_stage_type = two_stage_type
        if two_stage_type == "standard":
            self.enc_output = nn.Linear(d_model, d_model)
            self.enc_output_norm = LayerNorm(d_model)
            self.two_stage_wh_embedding = None
        if two_stage_type == "no":
            self.init_ref_points(num_queries)
        self.enc_out_class_embed = None
        self.enc_out_bbox_embed = None
        self._reset_parameters()

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in self.modules():
            if isinstance(m, MSDeformAttn):
                m._reset_parameters()
        if self.num_feature_levels > 1 and self.level_embed is not None:
            nn.init.normal_(self.level_embed)

    @staticmethod
    def get_valid_ratio(mask):
        _, H, W = mask.shape
        valid_H = torch.sum(~mask[:, :, 0], 1)
        valid_W = torch.sum(~mask[:, 0, :], 1)
        return torch.stack([valid_W.float() / W, valid_H.float() / H], -1)

    def init_ref_points(self, use_num_queries):
        self.refpoint_embed = nn.Embedding(use_num_queries, 4)

    def _level_tables(self, shapes, device):
        """Device copies of spatial_shapes / level_start_index, built once per level geometry
        (the reference uploads them on every forward; a cached tensor also keeps the forward
        capturable into a hipGraph)."""
        from . import geometry
        return geometry.level_tables(shapes, device)

    def prepare_inputs(self, srcs, masks, pos_embeds):
        """Flatten the levels: (src [B,S,C], mask [B,S], pos+level_embed [B,S,C], shapes list,
        spatial_shapes / level_start_index device tables, valid_ratios) -- reference :239-267."""
        src_flatten, mask_flatten, lvl_pos_embed_flatten, shapes = [], [], [], []
        for lvl, (src, mask, pos_embed) in enumerate(zip(srcs, masks, pos_embeds)):
            bs, c, h, w = src.shape
            shapes.append((h, w))
            pos_embed = pos_embed.flatten(2).transpose(1, 2)
            if self.num_feature_levels > 1 and self.level_embed is not None:
                pos_embed = pos_embed + self.level_embed[lvl].view(1, 1, -1)
            lvl_pos_embed_flatten.append(pos_embed)
            src_flatten.append(src.flatten(2).transpose(1, 2))
            mask_flatten.append(mask.flatten(1))
        src_flatten = torch.cat(src_flatten, 1)
        mask_flatten = torch.cat(mask_flatten, 1)
        lvl_pos_embed_flatten = torch.cat(lvl_pos_embed_flatten, 1)
        # the op wants the level table on the device (int64); the host copy `shapes` is kept for
        # everything that only needs Python ints, so nothing reads the device tensor back
        spatial_shapes, level_start_index = self._level_tables(tuple(shapes), src_flatten.device)
        from . import geometry
        if Switches.native_geometry and geometry.supported(mask_flatten, shapes):
            valid_ratios = geometry.valid_ratios(mask_flatten, shapes)   # one launch instead of 36, bit-identical
        else:
            valid_ratios = torch.stack([self.get_valid_ratio(m) for m in masks], 1)
        return (src_flatten, mask_flatten, lvl_pos_embed_flatten, shapes, spatial_shapes,
                level_start_index, valid_ratios)

    def select_queries(self, memory, mask_flatten, shapes, text_dict, refpoint_embed=None, tgt=None,
                       sort_for_topk=False):
        """Two-stage query selection: top-k pixels by max token logit (reference :301-372).
        -> (refpoint_embed, tgt, init_box_proposal, hs_enc, ref_enc).
        ``sort_for_topk``: take the first k of a stable descending sort instead of torch.topk (same
        indices unless logits tie); torch.topk inside a replayed hipGraph faults on ROCm 7.2."""
        bs = memory.shape[0]
        if self.two_stage_type == "standard":
            output_memory, output_proposals = gen_encoder_output_proposals(memory, mask_flatten, shapes)
            output_memory = self.enc_output_norm(self.enc_output(output_memory))
            enc_outputs_class_unselected = self.enc_out_class_embed(output_memory, text_dict)
            topk_logits = enc_outputs_class_unselected.max(-1)[0]
            enc_outputs_coord_unselected = self.enc_out_bbox_embed(output_memory) + output_proposals
            if sort_for_topk or Switches.sort_for_topk:
                topk_proposals = torch.sort(topk_logits, dim=1, descending=True, stable=True)[1][:, :self.num_queries]
            else:
                topk_proposals = torch.topk(topk_logits, self.num_queries, dim=1)[1]  # bs, nq (int64)
            gather4 = topk_proposals.unsqueeze(-1).repeat(1, 1, 4)
            refpoint_embed_undetach = torch.gather(enc_outputs_coord_unselected, 1, gather4)
            refpoint_embed_ = refpoint_embed_undetach.detach()
            init_box_proposal = torch.gather(output_proposals, 1, gather4).sigmoid()
            tgt_undetach = torch.gather(output_memory, 1,
                                        topk_proposals.unsqueeze(-1).repeat(1, 1, self.d_model))
            if self.embed_init_tgt:
                tgt_ = self.tgt_embed.weight[:, None, :].repeat(1, bs, 1).transpose(0, 1)
            else:
                tgt_ = tgt_undetach.detach()
            hs_enc = tgt_undetach.unsqueeze(0)
            ref_enc = refpoint_embed_undetach.sigmoid().unsqueeze(0)
        else:
            tgt_ = self.tgt_embed.weight[:, None, :].repeat(1, bs, 1).transpose(0, 1)
            refpoint_embed_ = self.refpoint_embed.weight[:, None, :].repeat(1, bs, 1).transpose(0, 1)
            init_box_proposal = refpoint_embed_.sigmoid()
            topk_proposals = hs_enc = ref_enc = None
        if refpoint_embed is not None:
            refpoint_embed = torch.cat([refpoint_embed, refpoint_embed_], dim=1)
            tgt = torch.cat([tgt, tgt_], dim=1)
        else:
            refpoint_embed, tgt = refpoint_embed_, tgt_
        self.last_topk_proposals = topk_proposals  # exposed for the bit-exact index parity tests
        return refpoint_embed, tgt, init_box_proposal, hs_enc, ref_enc

    def run_decoder(self, tgt, refpoint_embed, memory, mask_flatten, lvl_pos_embed_flatten, spatial_shapes,
                    level_start_index, valid_ratios, text_dict, attn_mask=None, no_padding=False):
        """The six decoder layers on the selected queries (reference :374-400) -> (hs, references)."""
        hs, references, _ = self.decoder(
            tgt=tgt.transpose(0, 1), memory=memory.transpose(0, 1),
            memory_key_padding_mask=None if no_padding else mask_flatten,
            pos=lvl_pos_embed_flatten.transpose(0, 1),
            refpoints_unsigmoid=refpoint_embed.transpose(0, 1), level_start_index=level_start_index,
            spatial_shapes=spatial_shapes, valid_ratios=valid_ratios, tgt_mask=attn_mask,
            memory_text=text_dict["encoded_text"], text_attention_mask=~text_dict["text_token_mask"])
        return hs, references

    def select_and_decode(self, memory, mask_flatten, lvl_pos_embed_flatten, shapes, spatial_shapes,
                          level_start_index, valid_ratios, text_dict, refpoint_embed=None, tgt=None,
                          attn_mask=None, no_padding=False, sort_for_topk=False):
        """Two-stage query selection (top-k by max token logit) + decoder (reference :301-415)."""
        refpoint_embed, tgt, init_box_proposal, hs_enc, ref_enc = self.select_queries(
            memory, mask_flatten, shapes, text_dict, refpoint_embed, tgt, sort_for_topk=sort_for_topk)
        hs, references = self.run_decoder(tgt, refpoint_embed, memory, mask_flatten, lvl_pos_embed_flatten,
                                          spatial_shapes, level_start_index, valid_ratios, text_dict,
                                          attn_mask, no_padding)
        return hs, references, hs_enc, ref_enc, init_box_proposal

    def fire_after_encoder(self):
        """Call (once) what ``after_encoder`` holds: a trainer puts the next minibatch's frozen front end there, so that it is
        queued when the encoder's forward has been launched and runs beside the launch-bound part of the step -- query
        selection, decoder, criterion, decoder backward -- instead of beside the backward's large GEMMs."""
        hook, self.__dict__["after_encoder"] = self.__dict__.get("after_encoder"), None
        if hook is not None:
            hook()

    def forward(self, srcs, masks, refpoint_embed, pos_embeds, tgt, attn_mask=None, text_dict=None,
                no_padding=False):
        """``no_padding``: the caller knows (on the host, from the image sizes) that ``masks`` are
        all False; the key-padding fills of the 12 MSDA calls and 6 fusion blocks (a pass over the
        45 MB value tensor each, forward and backward) are then skipped -- same results."""
        (src_flatten, mask_flatten, lvl_pos_embed_flatten, shapes, spatial_shapes, level_start_index,
         valid_ratios) = self.prepare_inputs(srcs, masks, pos_embeds)
        memory, memory_text, adapter_loss1 = self.encoder(
            src_flatten, pos=lvl_pos_embed_flatten, level_start_index=level_start_index,
            spatial_shapes=spatial_shapes, valid_ratios=valid_ratios,
            key_padding_mask=None if no_padding else mask_flatten,
            memory_text=text_dict["encoded_text"], text_attention_mask=~text_dict["text_token_mask"],
            position_ids=text_dict["position_ids"],
            text_self_attention_masks=text_dict["text_self_attention_masks"],
            spatial_shapes_list=shapes)
        text_dict["encoded_text"] = memory_text
        self.fire_after_encoder()
        hs, references, hs_enc, ref_enc, init_box_proposal = self.select_and_decode(
            memory, mask_flatten, lvl_pos_embed_flatten, shapes, spatial_shapes, level_start_index,
            valid_ratios, text_dict, refpoint_embed, tgt, attn_mask, no_padding=no_padding)
        return hs, references, hs_enc, ref_enc, init_box_proposal, adapter_loss1


def build_transformer(args):
    return Transformer(
        d_model=args.hidden_dim, dropout=args.dropout, nhead=args.nheads,
        num_queries=args.num_queries, dim_feedforward=args.dim_feedforward,
        num_encoder_layers=args.enc_layers, num_decoder_layers=args.dec_layers,
        normalize_before=args.pre_norm, return_intermediate_dec=True, query_dim=args.query_dim,
        activation=args.transformer_activation, num_patterns=args.num_patterns,
        num_feature_levels=args.num_feature_levels, enc_n_points=args.enc_n_points,
        dec_n_points=args.dec_n_points, learnable_tgt_init=True, two_stage_type=args.two_stage_type,
        embed_init_tgt=args.embed_init_tgt, use_text_enhancer=args.use_text_enhancer,
        use_fusion_layer=args.use_fusion_layer, use_checkpoint=args.use_checkpoint,
        use_transformer_ckpt=args.use_transformer_ckpt,
        use_text_cross_attention=args.use_text_cross_attention, text_dropout=args.text_dropout,
        fusion_dropout=args.fusion_dropout, fusion_droppath=args.fusion_droppath,
        use_adapter=getattr(args, "use_adapter", False))
