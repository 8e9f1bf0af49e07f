"""The deformable self-attention sublayer of an encoder layer with frozen weights as ONE autograd node
(reference transformer_for_adapter.py:888-899: ``src2 = self_attn(with_pos_embed(src, pos), reference_points, src, ...)``,
``src = norm1(src + dropout1(src2))``; module internals ms_deform_attn.py:286-340).

The image tokens ``src`` ([B, 22223, 256]: 45 MB at the bench size) are read three times -- as values, as queries and through
the residual connection -- so autograd, given the modules, forms three gradients for them and adds them in two passes over
the 45 MB tensor.  ``_SharedSourceProjections`` already lets the second projection's product accumulate into the first's;
here the residual path joins as well: the LayerNorm input gradient ``gs`` is the addend (beta = 1) of the first product,

    gs = LNbwd(g);  go = gs Wo;  gv, gloc, gattn = MSDA'(go);  gproj = sampling'(gloc, gattn)
    gsrc = gs + gv Wv + gproj Wq            (addmm_ into gs, twice)

The forward is the module's own sequence of launches (two projections, sampling locations + softmax, the MSDA op, output
projection, residual add + LayerNorm in one kernel).  fp32 GPU calls without padding mask only; otherwise the modules run."""
import torch
from torch.autograd.function import once_differentiable

from . import _C, _lib
from . import gemm_bf16x3 as g3


def applies(layer, src, pos, reference_points, spatial_shapes, key_padding_mask) -> bool:
    from .ms_deform_attn import _frozen_fp32_linear, _sampling_plan_ok
    ms, norm = layer.self_attn, layer.norm1
    if key_padding_mask is not None or pos is None or not src.is_cuda or src.dtype != torch.float32 or src.dim() != 3:
        return False
    if torch.is_autocast_enabled("cuda") or pos.requires_grad or (layer.training and layer.dropout1.p > 0.0):
        return False
    if not (ms.batch_first and ms.fuse_sampling_plan and _frozen_fp32_linear(ms.value_proj, src)
            and _frozen_fp32_linear(ms.output_proj, src)):
        return False
    if norm.weight is None or norm.bias is None or norm.weight.requires_grad or norm.bias.requires_grad:
        return False
    # the same LayerNorm kernel as the module path would take (dense.add_layer_norm: small inputs stay with ATen there, and the
    # padded / unpadded runs of one model must agree bit for bit)
    from .dense import LayerNorm, layer_norm_supported
    if not (isinstance(norm, LayerNorm) and norm.fused and LayerNorm.fused_residual and LayerNorm.fused_backward
            and layer_norm_supported(src, tuple(norm.normalized_shape), norm.weight, norm.bias)):
        return False
    C = src.shape[-1]
    if tuple(norm.normalized_shape) != (C,) or C % 4 or C > 1024 or ms.embed_dim != C:
        return False
    if ms._fused_query_projection() is None:
        return False
    probe = src.new_empty(0)
    return _sampling_plan_ok(probe, reference_points, spatial_shapes, ms.num_levels, ms.num_points)


class _FrozenEncoderAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, layer, src, pos, ref, shapes, level_start):
        lib = _lib.load()
        ms, norm = layer.self_attn, layer.norm1
        wq, bq = ms._fused_query_projection()
        wv, bv, wo, bo = ms.value_proj.weight, ms.value_proj.bias, ms.output_proj.weight, ms.output_proj.bias
        B, S, C = src.shape
        M, L, P = ms.num_heads, ms.num_levels, ms.num_points
        dev = src.device
        src = src.contiguous()
        ref = ref.contiguous()
        R = ref.shape[-1]
        rows = B * S
        f32 = dict(dtype=torch.float32, device=dev)
        s2 = src.view(rows, C)
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            arith = g3.enabled() and g3.supported(s2, C, C) and g3.supported(s2, wq.shape[0], C)
            if arith:   # the 256-wide products on the matrix cores in split arithmetic (gemm_bf16x3.py); the position code is
                #         added to the query inside the kernel where the panel kernel takes the product
                value = g3.linear(ms, "value", s2, wv, bv).view(B, S, M, C // M)
                proj = g3.linear(ms, "query", s2, wq, bq, add=pos.contiguous().view(rows, C))
            else:
                sp = (src + pos).view(rows, C)
                value = torch.addmm(bv, s2, wv.t()).view(B, S, M, C // M)
                proj = torch.addmm(bq, sp, wq.t())
            nproj = proj.shape[1]
            loc = torch.empty((B, S, M, L, P, 2), **f32)
            attn = torch.empty((B, S, M, L, P), **f32)
            rc = lib.zira_msda_sampling_fwd_f32(proj.data_ptr(), nproj, ref.data_ptr(), R, shapes.data_ptr(), rows, M, L, P,
                                                loc.data_ptr(), attn.data_ptr(), st)
            if rc != 0:
                raise RuntimeError("zira_msda_sampling_fwd_f32 failed with code %d" % rc)
            o = _C.ms_deform_attn_forward(value, shapes, level_start, loc, attn, ms.im2col_step)
            y = g3.linear(ms, "output", o.view(rows, C), wo, bo) if arith else torch.addmm(bo, o.view(rows, C), wo.t())
            out, s = torch.empty_like(src), torch.empty_like(src)
            stats = torch.empty((2, rows), **f32)
            rc = lib.zira_add_layernorm_fwd_f32(s2.data_ptr(), y.data_ptr(), norm.weight.data_ptr(), norm.bias.data_ptr(), rows, C,
                                                float(norm.eps), s.data_ptr(), out.data_ptr(), stats[0].data_ptr(),
                                                stats[1].data_ptr(), st)
            if rc != 0:
                raise RuntimeError("zira_add_layernorm_fwd_f32 failed with code %d" % rc)
        ctx.layer = layer
        ctx.dims = (B, S, C, M, L, P, R, nproj)
        ctx.arith = arith
        ctx.save_for_backward(value, loc, attn, ref, s, stats, shapes, level_start, wq)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        value, loc, attn, ref, s, stats, shapes, level_start, wq = ctx.saved_tensors
        lib = _lib.load()
        ms, norm = ctx.layer.self_attn, ctx.layer.norm1
        B, S, C, M, L, P, R, nproj = ctx.dims
        rows = B * S
        dev = g.device
        g = g.contiguous()
        with torch.cuda.device(dev):
            st = torch.cuda.current_stream(dev).cuda_stream
            gs = torch.empty_like(s)
            rc = lib.zira_layernorm_bwd_f32(g.data_ptr(), s.data_ptr(), norm.weight.data_ptr(), stats[0].data_ptr(),
                                            stats[1].data_ptr(), rows, C, gs.data_ptr(), st)
            if rc != 0:
                raise RuntimeError("zira_layernorm_bwd_f32 failed with code %d" % rc)
            gs2 = gs.view(rows, C)
            if ctx.arith:
                go = g3.linear_input_grad(ms, "output", gs2, ms.output_proj.weight).view(B, S, C)
            else:
                go = (gs2 @ ms.output_proj.weight).view(B, S, C)
            gv, gloc, gattn = _C.ms_deform_attn_backward(value, shapes, level_start, loc, attn, go, ms.im2col_step)
            gproj = torch.empty((rows, nproj), dtype=torch.float32, device=dev)
            rc = lib.zira_msda_sampling_bwd_f32(gloc.data_ptr(), gattn.data_ptr(), attn.data_ptr(), ref.data_ptr(), R,
                                                shapes.data_ptr(), rows, M, L, P, gproj.data_ptr(), nproj, st)
            if rc != 0:
                raise RuntimeError("zira_msda_sampling_bwd_f32 failed with code %d" % rc)
            # the three gradients of src meet in the GEMMs: residual path as the addend, then the two projections
            # (in place: torch.addmm(x, ...) would first COPY x into its result -- a pass of its own; gs is this node's)
            if ctx.arith:
                gx = g3.linear_input_grad(ms, "value", gv.view(rows, C), ms.value_proj.weight, accumulate_into=gs2)
                g3.linear_input_grad(ms, "query", gproj, wq, accumulate_into=gx)
            else:
                gx = gs2.addmm_(gv.view(rows, C), ms.value_proj.weight)
                gx.addmm_(gproj, wq)
        return None, gx.view(B, S, C), None, None, None, None


def attention_sublayer(layer, src, pos, reference_points, spatial_shapes, level_start_index):
    """norm1(src + self_attn(src + pos, reference_points, src)) of a frozen encoder layer; call only when ``applies()``."""
    from .ms_deform_attn import _check_levels_cover_value
    _check_levels_cover_value(spatial_shapes, src.shape[1], level_start_index)
    return _FrozenEncoderAttention.apply(layer, src, pos, reference_points, spatial_shapes, level_start_index)
