"""One cross-modal decoder layer with frozen weights as ONE autograd node with a hand-written backward
(reference transformer_for_adapter.py:910-1073: self-attention -> text cross-attention -> MSDA cross-attention -> FFN, post-LN).

Written as PyTorch modules the layer is ~70 launches forward + backward on B x 900 rows (a few microseconds each whatever they
compute): the position-code adds, the residual adds, the LayerNorms and their gradients, the gradient sums autograd forms where
a tensor has several consumers, and the copies between the decoder's [queries, batch, C] order and the MSDA op's
[batch, queries, C] order.  Here every one of those rides in the prologue or epilogue of the GEMM next to it
(``rowgemm``, csrc/rowgemm.hip), and the backward is written down once:

  forward                                            backward (gradient g of the layer output)
  qkv  = [tgt + pos | tgt] Win^T                     ds4, gh = LNbwd(g), (ds4 W2) * (h > 0)
  o1   = attention(q, k, v)                          g3   = gh W1 + ds4                            (addmm, beta = 1)
  t1   = LN(tgt + o1 Wo^T)                           go3, ds3 = (LNbwd(g3)) Wout, LNbwd(g3)        -> MSDA backward, batch-first
  q2   = (t1 + pos) Wq^T ; kv2 = text Wkv^T          g2   = gproj Wsamp + ds3
  o2   = attention(q2, k2, v2, text mask)            go2, ds2 = LNbwd(g2) Wo2 ; attention backward
  t2   = LN(t1 + o2 Wo2^T)                           g1   = dq2 Wq + ds2 ; gtext = dkv2 Wkv
  proj = (t2 + pos) Wsamp^T   (batch-first)          go1, ds1 = LNbwd(g1) Wo ; attention backward into [dq | dk | dv]
  o3   = MSDA(value, sampling(proj, ref))            gtgt = [dq | dk | dv] Win + ds1
  t3   = LN(t2 + o3 Wout^T)
  out  = LN(t3 + relu(t3 W1^T) W2^T)

Weights the forward needs transposed ([K, N]: a wave reads whole cache lines of them) are cached per layer and refreshed
in place when a parameter changes (captured hipGraphs keep reading the same buffers).  fp32 only; there is no fallback
inside: ``applies()`` says whether the layer can take this path, otherwise the module composition runs."""
import math
from typing import Optional

import torch
from torch import Tensor
from torch.autograd.function import once_differentiable

from . import _C, _lib
from .rowgemm import rowgemm


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


class LayerWeights:
    """Transposed / concatenated copies of a decoder layer's frozen weights, in the layouts the forward reads."""

    def __init__(self, layer):
        self.key = None
        self.refresh(layer)

    @staticmethod
    def _sources(layer):
        sa, ca, ms = layer.self_attn, layer.ca_text, layer.cross_attn
        return [sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias,
                ca.in_proj_weight, ca.in_proj_bias, ca.out_proj.weight, ca.out_proj.bias,
                ms.sampling_offsets.weight, ms.sampling_offsets.bias, ms.attention_weights.weight, ms.attention_weights.bias,
                ms.output_proj.weight, ms.output_proj.bias, layer.linear1.weight, layer.linear1.bias,
                layer.linear2.weight, layer.linear2.bias]

    def refresh(self, layer):
        src = self._sources(layer)
        key = tuple((p.data_ptr(), p._version) for p in src)
        if key == self.key:
            return
        sa, ca, ms = layer.self_attn, layer.ca_text, layer.cross_attn
        E = sa.embed_dim
        with torch.no_grad():
            new = {
                "sa_in_t": sa.in_proj_weight.t(),                       # [E, 3E]: q | k | v columns
                "sa_out_t": sa.out_proj.weight.t(),
                "ca_q_t": ca.in_proj_weight[:E].t(),
                "ca_kv_t": ca.in_proj_weight[E:].t(),                   # [E, 2E]: k | v columns
                "ca_out_t": ca.out_proj.weight.t(),
                "samp_w": torch.cat([ms.sampling_offsets.weight, ms.attention_weights.weight], 0),   # [3MLP, E] as stored
                "samp_b": torch.cat([ms.sampling_offsets.bias, ms.attention_weights.bias], 0),
                "out_t": ms.output_proj.weight.t(),
            }
            new["samp_t"] = new["samp_w"].t()
            dev = sa.in_proj_weight.device
            if self.key is None or self.sa_in_t.device != dev:   # first use, or the layer moved (model.to(...))
                for k, v in new.items():
                    setattr(self, k, v.contiguous().clone())
            else:   # in place: captured graphs read these buffers
                for k, v in new.items():
                    getattr(self, k).copy_(v)
        self.key = key


def _weights(layer) -> LayerWeights:
    w = getattr(layer, "_native_weights", None)
    if w is None:
        w = layer._native_weights = LayerWeights(layer)
    else:
        w.refresh(layer)
    return w


def applies(layer, tgt, pos, ref, text_lb, memory_value, self_attn_mask, tgt_key_padding_mask) -> bool:
    """Whether ``decoder_layer_forward`` can run this call: a CUDA fp32 call outside autocast on a layer with all weights
    frozen, heads of width 32, model width 256, no dropout in effect and no masks on the queries."""
    if not (tgt.is_cuda and tgt.dtype == torch.float32 and not torch.is_autocast_enabled("cuda")):
        return False
    if layer.self_attn is None or not layer.use_text_cross_attention or memory_value is None or pos is None:
        return False
    if self_attn_mask is not None or tgt_key_padding_mask is not None or pos.requires_grad or ref.requires_grad:
        return False
    if any(p.requires_grad for p in layer.parameters()) or any(p.dtype != torch.float32 for p in layer.parameters()):
        return False
    sa, ca, ms = layer.self_attn, layer.ca_text, layer.cross_attn
    E = tgt.shape[-1]
    if E != 256 or sa.num_heads * 32 != E or ca.num_heads * 32 != E or sa.in_proj_weight is None or ca.in_proj_weight is None:
        return False
    if layer.training and any(getattr(d, "p", 0.0) > 0.0 for d in (layer.dropout1, layer.dropout2, layer.dropout3, layer.dropout4,
                                                                  layer.catext_dropout)):
        return False
    if layer.training and (sa.dropout > 0.0 or ca.dropout > 0.0):
        return False
    if layer.activation is not torch.nn.functional.relu or layer.linear1.out_features % 128 or layer.linear1.out_features > 2048:
        return False
    mlp = ms.num_heads * ms.num_levels * ms.num_points
    if (3 * mlp) % 128 or ms.embed_dim != E or (ms.num_levels * ms.num_points) & (ms.num_levels * ms.num_points - 1):
        return False
    if memory_value.dtype != torch.float32 or memory_value.dim() not in (3, 4) or not memory_value.is_contiguous():
        return False
    return all(n.weight is not None and n.bias is not None for n in (layer.norm1, layer.norm2, layer.norm3, layer.catext_norm))


def _ln(norm):
    return (norm.weight, norm.bias, norm.eps)


class _FrozenDecoderLayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, layer, tgt, pos, ref_bf, text_lb, text_mask, value, shapes, level_start):
        lib = _lib.load()
        w = _weights(layer)
        sa, ca, ms = layer.self_attn, layer.ca_text, layer.cross_attn
        dev = tgt.device
        Q, B, E = tgt.shape
        T = text_lb.shape[0]
        H = sa.num_heads
        Mh, L, P = ms.num_heads, ms.num_levels, ms.num_points
        R = ref_bf.shape[-1]
        scale = 1.0 / math.sqrt(32.0)
        tgt = tgt.contiguous()
        pos = pos.contiguous()
        text_lb = text_lb.contiguous()
        st = _stream(dev)
        f32 = dict(dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            # ---- self-attention ----
            qkv = rowgemm(tgt, w.sa_in_t, w_is_nk=False, bias=sa.in_proj_bias, pos=pos, pos_cols=2 * E)       # [QB, 3E]
            o1 = torch.empty((Q * B, E), **f32)
            lse1 = torch.empty((B, H, Q), **f32)
            rc = lib.zira_attn_fwd_f32(qkv.data_ptr(), qkv.data_ptr() + 4 * E, qkv.data_ptr() + 8 * E, None, Q, Q, B, H, 32,
                                       3 * E, 3 * E, 3 * E, scale, o1.data_ptr(), lse1.data_ptr(), st)
            if rc != 0:
                raise RuntimeError("zira_attn_fwd_f32 failed with code %d" % rc)
            t1, s1, m1, r1 = rowgemm(o1, w.sa_out_t, w_is_nk=False, bias=sa.out_proj.bias, res=tgt, ln=_ln(layer.norm2), ln_save=True)
            # ---- text cross-attention ----
            q2 = rowgemm(t1, w.ca_q_t, w_is_nk=False, bias=ca.in_proj_bias[:E], pos=pos)
            kv2 = rowgemm(text_lb, w.ca_kv_t, w_is_nk=False, bias=ca.in_proj_bias[E:])                       # [TB, 2E]
            o2 = torch.empty((Q * B, E), **f32)
            lse2 = torch.empty((B, H, Q), **f32)
            rc = lib.zira_attn_fwd_f32(q2.data_ptr(), kv2.data_ptr(), kv2.data_ptr() + 4 * E,
                                       text_mask.data_ptr() if text_mask is not None else None, Q, T, B, H, 32, E, 2 * E, 2 * E,
                                       scale, o2.data_ptr(), lse2.data_ptr(), st)
            if rc != 0:
                raise RuntimeError("zira_attn_fwd_f32 failed with code %d" % rc)
            t2, s2, m2, r2 = rowgemm(o2, w.ca_out_t, w_is_nk=False, bias=ca.out_proj.bias, res=t1, ln=_ln(layer.catext_norm),
                                     ln_save=True)
            # ---- MSDA cross-attention (the op's side is batch-first) ----
            nproj = 3 * Mh * L * P
            proj = rowgemm(t2, w.samp_t, w_is_nk=False, bias=w.samp_b, pos=pos, batch=B, c_batch_first=True)  # [B*Q, 3MLP]
            loc = torch.empty((B, Q, Mh, L, P, 2), **f32)
            attn = torch.empty((B, Q, Mh, L, P), **f32)
            rc = lib.zira_msda_sampling_fwd_f32(proj.data_ptr(), nproj, ref_bf.data_ptr(), R, shapes.data_ptr(), B * Q, Mh, L, P,
                                                loc.data_ptr(), attn.data_ptr(), st)
            if rc != 0:
                raise RuntimeError("zira_msda_sampling_fwd_f32 failed with code %d" % rc)
            plan = None
            need_grad = any(ctx.needs_input_grad)
            if need_grad and _C.plan_applies(value, shapes, level_start, loc, ms.im2col_step):
                o3, plan = _C.ms_deform_attn_forward_plan(value, shapes, level_start, loc, attn, ms.im2col_step)
            else:
                o3 = _C.ms_deform_attn_forward(value, shapes, level_start, loc, attn, ms.im2col_step)
            t3, s3, m3, r3 = rowgemm(o3.view(B * Q, E), w.out_t, w_is_nk=False, bias=ms.output_proj.bias, res=t2, ln=_ln(layer.norm1),
                                     ln_save=True, batch=B, a_batch_first=True)
            # ---- FFN (the library's kernels win at 2048 columns) ----
            h = torch._addmm_activation(layer.linear1.bias, t3, layer.linear1.weight.t())
            y = torch.addmm(layer.linear2.bias, h, layer.linear2.weight.t())
            out, s4 = torch.empty_like(t3), torch.empty_like(t3)
            stats4 = torch.empty((2, Q * B), **f32)
            n3 = layer.norm3
            rc = lib.zira_add_layernorm_fwd_f32(t3.data_ptr(), y.data_ptr(), n3.weight.data_ptr(), n3.bias.data_ptr(), Q * B, E,
                                                float(n3.eps), s4.data_ptr(), out.data_ptr(), stats4[0].data_ptr(),
                                                stats4[1].data_ptr(), st)
            if rc != 0:
                raise RuntimeError("zira_add_layernorm_fwd_f32 failed with code %d" % rc)
        ctx.layer = layer
        ctx.plan = plan
        ctx.dims = (Q, B, E, T, H, Mh, L, P, R)
        ctx.save_for_backward(qkv, o1, lse1, s1, m1, r1, q2, kv2, o2, lse2, s2, m2, r2, attn, loc, s3, m3, r3, h, s4, stats4,
                              ref_bf, text_mask, value, shapes, level_start)
        return out.view(Q, B, E)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (qkv, o1, lse1, s1, m1, r1, q2, kv2, o2, lse2, s2, m2, r2, attn, loc, s3, m3, r3, h, s4, stats4,
         ref_bf, text_mask, value, shapes, level_start) = ctx.saved_tensors
        layer = ctx.layer
        lib = _lib.load()
        sa, ca, ms = layer.self_attn, layer.ca_text, layer.cross_attn
        w = layer._native_weights
        Q, B, E, T, H, Mh, L, P, R = ctx.dims
        dev = g.device
        st = _stream(dev)
        f32 = dict(dtype=torch.float32, device=dev)
        scale = 1.0 / math.sqrt(32.0)
        g = g.contiguous().view(Q * B, E)
        with torch.cuda.device(dev):
            # ---- FFN ----  (LayerNorm gradient, product with W2 and the ReLU gradient in one launch)
            gh, ds4 = rowgemm(g, layer.linear2.weight, w_is_nk=False, mask=h, lnb=(s4, layer.norm3.weight, stats4[0], stats4[1]),
                              lnb_save=True)
            g3 = ds4.addmm_(gh, layer.linear1.weight)   # (in place: ds4 is this node's)
            # ---- MSDA cross-attention ----
            go3, ds3 = rowgemm(g3, ms.output_proj.weight, w_is_nk=False, lnb=(s3, layer.norm1.weight, m3, r3), lnb_save=True,
                               batch=B, c_batch_first=True)
            kw = {} if ctx.plan is None else {"plan": ctx.plan}
            gvalue, gloc, gattn = _C.ms_deform_attn_backward(value, shapes, level_start, loc, attn, go3.view(B, Q, E),
                                                             ms.im2col_step, **kw)
            nproj = 3 * Mh * L * P
            gproj = torch.empty((B * Q, nproj), **f32)
            rc = lib.zira_msda_sampling_bwd_f32(gloc.data_ptr(), gattn.data_ptr(), attn.data_ptr(), ref_bf.data_ptr(), R,
                                                shapes.data_ptr(), B * Q, Mh, L, P, gproj.data_ptr(), nproj, st)
            if rc != 0:
                raise RuntimeError("zira_msda_sampling_bwd_f32 failed with code %d" % rc)
            g2 = rowgemm(gproj, w.samp_w, w_is_nk=False, res=ds3, batch=B, a_batch_first=True)
            # ---- text cross-attention ----
            go2, ds2 = rowgemm(g2, ca.out_proj.weight, w_is_nk=False, lnb=(s2, layer.catext_norm.weight, m2, r2), lnb_save=True)
            dq2 = torch.empty((Q * B, E), **f32)
            dkv2 = torch.empty((T * B, 2 * E), **f32)
            nscr = lib.zira_attn_bwd_scratch_floats(Q, T, B, H)
            scratch = torch.empty((nscr,), **f32)
            rc = lib.zira_attn_bwd_ld_f32(q2.data_ptr(), kv2.data_ptr(), kv2.data_ptr() + 4 * E,
                                          text_mask.data_ptr() if text_mask is not None else None, o2.data_ptr(), go2.data_ptr(),
                                          lse2.data_ptr(), Q, T, B, H, 32, E, 2 * E, 2 * E, scale, dq2.data_ptr(), dkv2.data_ptr(),
                                          dkv2.data_ptr() + 4 * E, E, 2 * E, 2 * E, scratch.data_ptr(), nscr, st)
            if rc != 0:
                raise RuntimeError("zira_attn_bwd_ld_f32 failed with code %d" % rc)
            gtext = None
            if ctx.needs_input_grad[4]:
                gtext = rowgemm(dkv2, ca.in_proj_weight[E:], w_is_nk=False).view(T, B, E)
            g1 = rowgemm(dq2, ca.in_proj_weight[:E], w_is_nk=False, res=ds2)
            # ---- self-attention ----
            go1, ds1 = rowgemm(g1, sa.out_proj.weight, w_is_nk=False, lnb=(s1, layer.norm2.weight, m1, r1), lnb_save=True)
            dqkv = torch.empty((Q * B, 3 * E), **f32)
            nscr = lib.zira_attn_bwd_scratch_floats(Q, Q, B, H)
            scratch = torch.empty((nscr,), **f32)
            rc = lib.zira_attn_bwd_ld_f32(qkv.data_ptr(), qkv.data_ptr() + 4 * E, qkv.data_ptr() + 8 * E, None, o1.data_ptr(),
                                          go1.data_ptr(), lse1.data_ptr(), Q, Q, B, H, 32, 3 * E, 3 * E, 3 * E, scale,
                                          dqkv.data_ptr(), dqkv.data_ptr() + 4 * E, dqkv.data_ptr() + 8 * E, 3 * E, 3 * E, 3 * E,
                                          scratch.data_ptr(), nscr, st)
            if rc != 0:
                raise RuntimeError("zira_attn_bwd_ld_f32 failed with code %d" % rc)
            gtgt = None
            if ctx.needs_input_grad[1]:
                gtgt = rowgemm(dqkv, sa.in_proj_weight, w_is_nk=False, res=ds1).view(Q, B, E)
        return None, gtgt, None, None, gtext, None, (gvalue if ctx.needs_input_grad[6] else None), None, None


def decoder_layer_forward(layer, tgt: Tensor, pos: Tensor, ref: Tensor, text_lb: Tensor, text_mask: Optional[Tensor],
                          memory_value: Tensor, shapes: Tensor, level_start: Tensor, ref_bf: Optional[Tensor] = None) -> Tensor:
    """The layer's output [Q, B, C] for ``tgt`` / ``pos`` [Q, B, C], reference points ``ref`` [Q, B, L, 2 | 4], text memory
    ``text_lb`` [T, B, C] with additive key mask ``text_mask`` [B, T] (or None) and ``memory_value`` = the layer's
    value_proj(memory) [B, S, M, D] with padded rows zeroed.  Call only when ``applies()`` said so."""
    from .ms_deform_attn import _check_levels_cover_value
    ms = layer.cross_attn
    B, S = memory_value.shape[:2]
    _check_levels_cover_value(shapes, S, level_start)
    memory_value = memory_value.view(B, S, ms.num_heads, -1)
    if ref_bf is None:   # (``ref_bf``: the reference points batch-first, when the caller has them: prep_queries)
        ref_bf = ref.transpose(0, 1).contiguous()
    return _FrozenDecoderLayer.apply(layer, tgt, pos, ref_bf, text_lb, text_mask, memory_value, shapes, level_start)


# ---- between the layers: iterative box refinement + the LayerNorm of the intermediate output --------------------------------
# (reference TransformerDecoder.forward, transformer_for_adapter.py:700-806: per layer  delta = bbox_embed(output);
#  new_ref = sigmoid(delta + inverse_sigmoid(ref));  intermediate.append(norm(output)) -- as PyTorch ops 14 launches forward
#  and 8 backward on 1800 rows, plus the add autograd needs because `output` feeds both.)

class MLPWeights:
    """Transposed copies of the first two layers of a frozen box MLP (the forward's rowgemm wants [K, N])."""

    def __init__(self, mlp):
        self.key = None
        self.refresh(mlp)

    def refresh(self, mlp):
        l0, l1 = mlp.layers[0], mlp.layers[1]
        key = tuple((p.data_ptr(), p._version) for p in (l0.weight, l1.weight))
        if key == self.key:
            return
        with torch.no_grad():
            if self.key is None or self.w0_t.device != l0.weight.device:   # first use, or the module moved
                self.w0_t, self.w1_t = l0.weight.t().contiguous().clone(), l1.weight.t().contiguous().clone()
            else:   # in place: captured graphs read these buffers
                self.w0_t.copy_(l0.weight.t())
                self.w1_t.copy_(l1.weight.t())
        self.key = key


def _mlp_weights(mlp) -> MLPWeights:
    w = getattr(mlp, "_native_weights", None)
    if w is None:
        w = mlp._native_weights = MLPWeights(mlp)
    else:
        w.refresh(mlp)
    return w


def refine_applies(decoder, layer_id, output, ref) -> bool:
    """Whether ``refine_and_norm`` can run: fp32 GPU call outside autocast, frozen 256 -> 256 -> 256 -> 4 box MLP with
    biases, frozen affine LayerNorm, 4-d reference boxes that take no part in autograd."""
    if not (output.is_cuda and output.dtype == torch.float32 and not torch.is_autocast_enabled("cuda")):
        return False
    if decoder.bbox_embed is None or decoder.norm is None or ref.shape[-1] != 4 or ref.requires_grad:
        return False
    mlp, norm = decoder.bbox_embed[layer_id], decoder.norm
    layers = getattr(mlp, "layers", None)
    if layers is None or len(layers) != 3 or output.shape[-1] != 256:
        return False
    if [tuple(l.weight.shape) for l in layers] != [(256, 256), (256, 256), (4, 256)] or any(l.bias is None for l in layers):
        return False
    params = [p for l in layers for p in (l.weight, l.bias)] + [norm.weight, norm.bias]
    if any(p is None or p.requires_grad or p.dtype != torch.float32 for p in params):
        return False
    return tuple(norm.normalized_shape) == (256,)


class _RefineAndNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, decoder, layer_id, output, ref):
        lib = _lib.load()
        mlp, norm = decoder.bbox_embed[layer_id], decoder.norm
        w = _mlp_weights(mlp)
        l0, l1, l2 = mlp.layers
        Q, B, E = output.shape
        rows = Q * B
        dev = output.device
        output = output.contiguous()
        ref = ref.contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            st = _stream(dev)
            h1 = rowgemm(output, w.w0_t, w_is_nk=False, bias=l0.bias, relu=True)
            h2 = rowgemm(h1, w.w1_t, w_is_nk=False, bias=l1.bias, relu=True)
            new_ref = torch.empty((Q, B, 4), **f32)
            rc = lib.zira_box_refine_fwd_f32(h2.data_ptr(), l2.weight.data_ptr(), l2.bias.data_ptr(), ref.data_ptr(), rows, E, 1e-3,
                                             new_ref.data_ptr(), st)
            if rc != 0:
                raise RuntimeError("zira_box_refine_fwd_f32 failed with code %d" % rc)
            normed = torch.empty_like(output)
            stats = torch.empty((2, rows), **f32)
            rc = lib.zira_layernorm_fwd_f32(output.data_ptr(), norm.weight.data_ptr(), norm.bias.data_ptr(), rows, E, float(norm.eps),
                                            normed.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), st)
            if rc != 0:
                raise RuntimeError("zira_layernorm_fwd_f32 failed with code %d" % rc)
        ctx.mods = (mlp, norm)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(output, h1, h2, new_ref, stats)
        return new_ref, normed

    @staticmethod
    @once_differentiable
    def backward(ctx, g_new, g_normed):
        output, h1, h2, new_ref, stats = ctx.saved_tensors
        mlp, norm = ctx.mods
        l0, l1, l2 = mlp.layers
        lib = _lib.load()
        Q, B, E = output.shape
        rows = Q * B
        dev = output.device
        gx = None
        with torch.cuda.device(dev):
            st = _stream(dev)
            if g_normed is not None:
                g_normed = g_normed.contiguous()
                gx = torch.empty_like(output)
                rc = lib.zira_layernorm_bwd_f32(g_normed.data_ptr(), output.data_ptr(), norm.weight.data_ptr(), stats[0].data_ptr(),
                                                stats[1].data_ptr(), rows, E, gx.data_ptr(), st)
                if rc != 0:
                    raise RuntimeError("zira_layernorm_bwd_f32 failed with code %d" % rc)
            if g_new is not None:
                g_new = g_new.contiguous()
                gh2 = torch.empty_like(h2)
                rc = lib.zira_box_refine_bwd_f32(g_new.data_ptr(), new_ref.data_ptr(), l2.weight.data_ptr(), h2.data_ptr(), rows, E,
                                                 gh2.data_ptr(), st)
                if rc != 0:
                    raise RuntimeError("zira_box_refine_bwd_f32 failed with code %d" % rc)
                gh1 = rowgemm(gh2, l1.weight, w_is_nk=False, mask=h1)
                gx = rowgemm(gh1, l0.weight, w_is_nk=False, res=gx).view(Q, B, E)
        return None, None, gx, None


def refine_and_norm(decoder, layer_id, output: Tensor, ref: Tensor):
    """(sigmoid(bbox_embed[layer_id](output) + inverse_sigmoid(ref)), norm(output)) as one autograd node: 4 launches forward,
    4 backward, and the two gradients of ``output`` meet inside the last GEMM.  Call only when ``refine_applies()`` said so."""
    return _RefineAndNorm.apply(decoder, layer_id, output, ref)


# ---- in front of a layer: the boxes scaled per level, their sine embedding and the query position code -----------------------

def prep_applies(decoder, reference_points, valid_ratios) -> bool:
    """Whether ``prep_queries`` can run: fp32 GPU boxes [Q, B, 4] outside autograd and autocast, frozen 512 -> 256 -> 256
    position MLP with biases, no query scale."""
    from .utils import NATIVE_REFPOINT_OPS
    if not (NATIVE_REFPOINT_OPS and reference_points.is_cuda and reference_points.dtype == torch.float32
            and reference_points.dim() == 3 and reference_points.shape[-1] == 4 and not torch.is_autocast_enabled("cuda")):
        return False
    if reference_points.requires_grad or valid_ratios.requires_grad or valid_ratios.dtype != torch.float32:
        return False
    if decoder.query_scale is not None or decoder.query_pos_sine_scale is not None:
        return False
    layers = getattr(decoder.ref_point_head, "layers", None)
    if layers is None or len(layers) != 2 or [tuple(l.weight.shape) for l in layers] != [(256, 512), (256, 256)]:
        return False
    return all(l.bias is not None and not l.weight.requires_grad and not l.bias.requires_grad and l.weight.dtype == torch.float32
               for l in layers)


_DIM_T = {}


def prep_queries(decoder, reference_points: Tensor, valid_ratios: Tensor):
    """(reference_points_input [Q, B, L, 4], the same batch-first [B, Q, L, 4], query_sine_embed [Q, B, 512], query_pos [Q, B, 256])
    of a decoder layer (reference transformer_for_adapter.py:760-776) in three launches; no autograd (nothing here has a
    gradient when ``prep_applies()``)."""
    from .utils import _dim_t
    lib = _lib.load()
    dev = reference_points.device
    Q, B, _ = reference_points.shape
    L = valid_ratios.shape[1]
    dim_t = _DIM_T.get(str(dev))
    if dim_t is None:
        dim_t = _DIM_T[str(dev)] = _dim_t(128, 10000, dev)
    ref = reference_points.contiguous()
    ratio = valid_ratios.contiguous()
    f32 = dict(dtype=torch.float32, device=dev)
    ref_in = torch.empty((Q, B, L, 4), **f32)
    ref_bf = torch.empty((B, Q, L, 4), **f32)
    sine = torch.empty((Q, B, 512), **f32)
    mlp = decoder.ref_point_head
    w = _mlp_weights(mlp)
    with torch.cuda.device(dev):
        rc = lib.zira_decoder_prep_f32(ref.data_ptr(), ratio.data_ptr(), dim_t.data_ptr(), Q, B, L, 128, 2 * math.pi,
                                       ref_in.data_ptr(), ref_bf.data_ptr(), sine.data_ptr(), _stream(dev))
        if rc != 0:
            raise RuntimeError("zira_decoder_prep_f32 failed with code %d" % rc)
        h = rowgemm(sine, w.w0_t, w_is_nk=False, bias=mlp.layers[0].bias, relu=True)
        query_pos = rowgemm(h, w.w1_t, w_is_nk=False, bias=mlp.layers[1].bias).view(Q, B, 256)
    return ref_in, ref_bf, sine, query_pos
