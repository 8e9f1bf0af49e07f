"""Host-side mirror of the reference op module (groundingdino/models/GroundingDINO/ms_deform_attn.py).

Same public names and call signatures:

* ``MultiScaleDeformableAttnFunction.apply(value, spatial_shapes, level_start_index,
  sampling_locations, attention_weights, im2col_step)``            (reference :38-87)
* ``MultiScaleDeformableAttention(embed_dim, num_heads, num_levels, num_points, img2col_step,
  batch_first)`` with parameters ``sampling_offsets / attention_weights / value_proj /
  output_proj`` and the same ``forward`` keyword arguments                 (reference :133-355)

* ``multi_scale_deformable_attn_pytorch(value, value_spatial_shapes, sampling_locations,
  attention_weights)``: the pure-PyTorch restatement for CPU tensors          (reference :90-130)

On GPU tensors the sampling + aggregation always runs in the gfx950 kernels behind ``_C`` and fails
loudly if the HIP library is missing -- the PyTorch path is only ever taken for CPU tensors, as in
the reference module (:326-348).  It is product code written from the op's arithmetic; the test
oracle under ``oracle/`` is a separate C restatement and is never imported here.
"""
import math
import warnings
import weakref
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _C


class MultiScaleDeformableAttnFunction(Function):
    """autograd glue around the two native entry points (reference ms_deform_attn.py:38-87)."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        # How the backward is cut into tiles depends on the sampling locations (and its records carry the attention weights): for sparse GPU calls (decoder
        # cross-attention) that plan is made here, in the forward's own launch, off the backward's critical path.
        ctx.plan = None
        if any(ctx.needs_input_grad) and _C.plan_applies(value, value_spatial_shapes, value_level_start_index,
                                                        sampling_locations, im2col_step):
            output, ctx.plan = _C.ms_deform_attn_forward_plan(value, value_spatial_shapes, value_level_start_index,
                                                             sampling_locations, attention_weights, im2col_step)
        else:
            output = _C.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                                               sampling_locations, attention_weights, im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, start, loc, attn = ctx.saved_tensors
        kw = {} if ctx.plan is None else {"plan": ctx.plan}
        grad_value, grad_loc, grad_attn = _C.ms_deform_attn_backward(
            value, shapes, start, loc, attn, grad_output.contiguous(), ctx.im2col_step, **kw)
        return grad_value, None, None, grad_loc, grad_attn, None


FUSED_LOCATIONS = True   # module-level switch for A/B runs


class _SamplingPlan(Function):
    """softmax of the attention logits + sampling locations from the ONE projection of the query (``[B, Q, 3*M*L*P]``:
    offsets, then logits), one native launch each way (csrc/sampling.hip); reference ms_deform_attn.py:290-325.
    The reference points get no gradient (callers check ``requires_grad``)."""

    @staticmethod
    def forward(ctx, proj, reference_points, spatial_shapes, M, L, P):
        from . import _lib
        lib = _lib.load()
        B, Q, ld = proj.shape
        proj = proj.contiguous()
        ref = reference_points.contiguous()
        R = ref.shape[-1]
        loc = torch.empty((B, Q, M, L, P, 2), dtype=torch.float32, device=proj.device)
        attn = torch.empty((B, Q, M, L, P), dtype=torch.float32, device=proj.device)
        with torch.cuda.device(proj.device):
            rc = lib.zira_msda_sampling_fwd_f32(proj.data_ptr(), ld, ref.data_ptr(), R, spatial_shapes.data_ptr(), B * Q, M, L, P,
                                                loc.data_ptr(), attn.data_ptr(), torch.cuda.current_stream(proj.device).cuda_stream)
        if rc != 0:
            raise RuntimeError("zira_msda_sampling_fwd_f32 failed with code %d" % rc)
        ctx.save_for_backward(attn, ref, spatial_shapes)
        ctx.dims = (B, Q, ld, M, L, P, R)
        return loc, attn

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_loc, grad_attn):
        from . import _lib
        lib = _lib.load()
        attn, ref, spatial_shapes = ctx.saved_tensors
        B, Q, ld, M, L, P, R = ctx.dims
        grad_loc, grad_attn = grad_loc.contiguous(), grad_attn.contiguous()
        grad_proj = torch.empty((B, Q, ld), dtype=torch.float32, device=attn.device)
        if ld != 3 * M * L * P:
            grad_proj.zero_()
        with torch.cuda.device(attn.device):
            rc = lib.zira_msda_sampling_bwd_f32(grad_loc.data_ptr(), grad_attn.data_ptr(), attn.data_ptr(), ref.data_ptr(), R,
                                                spatial_shapes.data_ptr(), B * Q, M, L, P, grad_proj.data_ptr(), ld,
                                                torch.cuda.current_stream(attn.device).cuda_stream)
        if rc != 0:
            raise RuntimeError("zira_msda_sampling_bwd_f32 failed with code %d" % rc)
        return grad_proj, None, None, None, None, None


class _SharedSourceProjections(Function):
    """value_proj(src) and the fused query projection of (src + pos) as ONE autograd node (frozen weights): the encoder's
    self-attention reads its image tokens three times -- as values, as queries and through the residual connection -- and
    autograd would form the two projections' input gradients separately and add them with a pass over the 45 MB tensor.
    Here the second product accumulates into the first (``addmm_``, beta = 1): one add less per layer and direction.
    Reference: ms_deform_attn.py:286-295 (value_proj, sampling_offsets / attention_weights on the same source)."""

    @staticmethod
    def forward(ctx, src, pos, wv, bv, wq, bq):
        c = src.shape[-1]
        s2 = src.reshape(-1, c)
        q2 = s2 if pos is None else (src + pos).reshape(-1, c)
        value = torch.addmm(bv, s2, wv.t()).view(*src.shape[:-1], wv.shape[0])
        proj = torch.addmm(bq, q2, wq.t()).view(*src.shape[:-1], wq.shape[0])
        ctx.save_for_backward(wv, wq)
        ctx.pos_grad = pos is not None and pos.requires_grad
        ctx.src_shape = src.shape
        return value, proj

    @staticmethod
    @once_differentiable
    def backward(ctx, g_value, g_proj):
        wv, wq = ctx.saved_tensors
        gx = None
        if g_value is not None:
            gx = g_value.reshape(-1, g_value.shape[-1]) @ wv
        g_pos = None
        if g_proj is not None:
            gp2 = g_proj.reshape(-1, g_proj.shape[-1])
            if ctx.pos_grad:
                g_pos = (gp2 @ wq).view(ctx.src_shape)
            gx = gp2 @ wq if gx is None else gx.addmm_(gp2, wq)
        return (None if gx is None else gx.view(ctx.src_shape)), g_pos, None, None, None, None


class _MultiValueProjections(Function):
    """The value projections of SEVERAL deformable-attention modules on one source (the six decoder layers all read the
    encoder's output, reference transformer_for_adapter.py:1059 inside the loop :700-806) as one autograd node, frozen
    weights: the input gradient is one buffer that the n products accumulate into, instead of n tensors of 45 MB and n - 1
    adds.  ``wb`` = n weights, then n biases."""

    @staticmethod
    def forward(ctx, x, n, *wb):
        from . import gemm_bf16x3 as g3
        ws, bs = wb[:n], wb[n:]
        x2 = x.reshape(-1, x.shape[-1])
        ctx.save_for_backward(*ws)
        ctx.x_shape = x.shape
        # (split-bf16 products on the bf16 matrix cores where asked for: gemm_bf16x3.py; the planes are cached on the weights' own
        #  parameters -- a Function has no module -- and follow them in place)
        ctx.arith = (g3.enabled() and x2.shape[0] >= 1024 and x2.is_contiguous()
                     and all(g3.supported(x2, w.shape[0], w.shape[1]) for w in ws))
        if ctx.arith:
            return tuple(g3.linear(w, "self", x2, w, b).view(*x.shape[:-1], w.shape[0]) for w, b in zip(ws, bs))
        return tuple(torch.addmm(b, x2, w.t()).view(*x.shape[:-1], w.shape[0]) for w, b in zip(ws, bs))

    @staticmethod
    @once_differentiable
    def backward(ctx, *gs):
        ws = ctx.saved_tensors
        gx = None
        for g, w in zip(gs, ws):
            if g is None:
                continue
            g2 = g.reshape(-1, g.shape[-1])
            if ctx.arith and g2.is_contiguous():
                from . import gemm_bf16x3 as g3
                gx = g3.linear_input_grad(w, "self", g2, w, accumulate_into=gx)
            else:
                gx = g2 @ w if gx is None else gx.addmm_(g2, w)
        return (None if gx is None else gx.view(ctx.x_shape), None) + (None,) * (2 * len(ws))


def _frozen_fp32_linear(lin, x):
    return (x.is_cuda and x.dtype == torch.float32 and lin.weight.dtype == torch.float32 and lin.bias is not None
            and not lin.weight.requires_grad and not lin.bias.requires_grad and not torch.is_autocast_enabled())


def multi_value_projections(modules, source, key_padding_mask=None):
    """``[m.value_proj(source) for m in modules]`` (padded positions zeroed, as ``MultiScaleDeformableAttention.forward``
    does), through one autograd node when every projection is frozen fp32 on the GPU; None otherwise (the modules then
    project for themselves).  Hand the i-th result to ``modules[i](..., value_projected=...)``."""
    if not modules or not all(_frozen_fp32_linear(m.value_proj, source) for m in modules) or not source.requires_grad:
        return None
    outs = _MultiValueProjections.apply(source, len(modules), *[m.value_proj.weight for m in modules],
                                        *[m.value_proj.bias for m in modules])
    if key_padding_mask is not None:
        outs = tuple(o.masked_fill(key_padding_mask[..., None], float(0)) for o in outs)
    return list(outs)


def _sampling_plan_ok(proj, reference_points, spatial_shapes, L, P):
    LP = L * P
    return (proj.is_cuda and proj.dtype == torch.float32 and reference_points.dtype == torch.float32
            and not reference_points.requires_grad and reference_points.shape[-1] in (2, 4)
            and LP <= 64 and (LP & (LP - 1)) == 0 and spatial_shapes.is_cuda and spatial_shapes.dtype == torch.int64
            and spatial_shapes.is_contiguous() and not torch.is_autocast_enabled())


def sampling_locations_from_reference_points(reference_points, sampling_offsets, spatial_shapes,
                                             num_points):
    """Sampling-location arithmetic of the reference module (ms_deform_attn.py:305-325).

    reference_points ``[B,Q,L,2]`` (encoder: pixel-centre grid) or ``[B,Q,L,4]`` (decoder:
    cx,cy,w,h boxes); sampling_offsets ``[B,Q,M,L,P,2]``; returns ``[B,Q,M,L,P,2]`` in (x, y).
    """
    last = reference_points.shape[-1]
    if last == 2:
        # offsets are in pixels of each level: normalise by (W, H)
        normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
        if sampling_offsets.is_floating_point() and FUSED_LOCATIONS:
            # same two roundings (quotient, sum) as the reference's expression, in one pass: the int64
            # divisor is converted once ([L, 2], exact) instead of inside a dtype-casting kernel over
            # the [B, Q, M, L, P, 2] offsets (43 us at the encoder shape, twice per layer and direction)
            return torch.addcdiv(reference_points[:, :, None, :, None, :], sampling_offsets,
                                 normalizer.to(sampling_offsets.dtype)[None, None, None, :, None, :])
        return (reference_points[:, :, None, :, None, :]
                + sampling_offsets / normalizer[None, None, None, :, None, :])
    if last == 4:
        # offsets are fractions of half the box size, spread over the P points
        return (reference_points[:, :, None, :, None, :2]
                + sampling_offsets / num_points * reference_points[:, :, None, :, None, 2:] * 0.5)
    raise ValueError(
        "Last dim of reference_points must be 2 or 4, but get {} instead.".format(last))


def multi_scale_deformable_attn_pytorch(value, value_spatial_shapes, sampling_locations,
                                        attention_weights):
    """Pure-PyTorch multi-scale deformable attention: the CPU path of the op (reference
    ms_deform_attn.py:90-130, the fallback its module takes for non-CUDA tensors :326-348).

    value ``[B,S,M,D]``, value_spatial_shapes ``[L,2]`` (H, W), sampling_locations
    ``[B,Q,M,L,P,2]`` (x, y in [0, 1]), attention_weights ``[B,Q,M,L,P]`` -> ``[B,Q,M*D]``.
    Per level: the head-major value map ``[B*M, D, H, W]`` is sampled bilinearly with zero padding
    at ``2 * loc - 1`` (``align_corners=False``, i.e. pixel coordinate = loc * size - 0.5: the
    arithmetic of the native kernels), then the L*P samples of a query are mixed with its attention
    weights.  Differentiable through autograd in all three inputs."""
    B, _, M, D = value.shape
    _, Q, _, L, P, _ = sampling_locations.shape
    sizes = [(int(h), int(w)) for h, w in value_spatial_shapes.tolist()]
    grids = 2 * sampling_locations - 1
    start, sampled = 0, []
    for lvl, (h, w) in enumerate(sizes):
        # [B, h*w, M, D] -> [B*M, D, h, w]
        v = value[:, start:start + h * w].permute(0, 2, 3, 1).reshape(B * M, D, h, w)
        start += h * w
        # [B, Q, M, P, 2] -> [B*M, Q, P, 2]
        g = grids[:, :, :, lvl].permute(0, 2, 1, 3, 4).reshape(B * M, Q, P, 2)
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
    # [B*M, D, Q, L*P] weighted by [B*M, 1, Q, L*P]
    weights = attention_weights.permute(0, 2, 1, 3, 4).reshape(B * M, 1, Q, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * weights).sum(-1)
    return out.view(B, M * D, Q).transpose(1, 2).contiguous()


class _VerifiedLevels:
    """Level tables whose sum(H*W) and level_start_index have been read back and checked once.
    Keyed on the tensor OBJECTS (weak references, plus their versions): a recycled storage address
    under a new tensor is a new entry, unlike a data_ptr() key."""

    def __init__(self):
        self._seen = {}

    def check(self, spatial_shapes, level_start_index, num_value):
        version = getattr(spatial_shapes, "_version", 0) if not spatial_shapes.is_inference() else 0
        key = (id(spatial_shapes), id(level_start_index), int(num_value))
        hit = self._seen.get(key)
        if hit is not None and hit[0]() is spatial_shapes and hit[2] == version and \
                (level_start_index is None or hit[1]() is level_start_index):
            return
        hw = spatial_shapes[:, 0] * spatial_shapes[:, 1]
        assert int(hw.sum()) == num_value, "spatial_shapes do not cover the value tokens"
        if level_start_index is not None:
            want = torch.cat([hw.new_zeros(1), hw.cumsum(0)[:-1]])
            assert torch.equal(level_start_index.to(want.dtype), want), \
                "level_start_index is not the running sum of H*W"
        if len(self._seen) > 256:
            self._seen.clear()
        self._seen[key] = (weakref.ref(spatial_shapes),
                           weakref.ref(level_start_index) if level_start_index is not None else None, version)


_VERIFIED_LEVELS = _VerifiedLevels()


def _check_levels_cover_value(spatial_shapes, num_value, level_start_index=None):
    """The reference asserts sum(H*W) == num_value on every call (ms_deform_attn.py:284), which
    reads a device tensor back (a host sync, 12x per step).  Same check here (plus
    level_start_index, the only guard of the kernels' row addressing), but a given pair of level
    tensors is only read back once."""
    _VERIFIED_LEVELS.check(spatial_shapes, level_start_index, num_value)


class MultiScaleDeformableAttention(nn.Module):
    """Multi-scale deformable attention (Deformable-DETR) with the reference's parameter names.

    State-dict keys: ``sampling_offsets.{weight,bias}`` (M*L*P*2 x C), ``attention_weights.*``
    (M*L*P x C), ``value_proj.*``, ``output_proj.*`` -- identical to the reference so its
    checkpoints load unchanged.
    """

    def __init__(self, embed_dim: int = 256, num_heads: int = 8, num_levels: int = 4,
                 num_points: int = 4, img2col_step: int = 64, batch_first: bool = False):
        super().__init__()
        if embed_dim % num_heads != 0:
            raise ValueError("embed_dim must be divisible by num_heads, but got {} and {}".format(
                embed_dim, num_heads))
        head_dim = embed_dim // num_heads
        if head_dim & (head_dim - 1):
            warnings.warn("head_dim=%d is not a power of two: the op takes its generic "
                          "(slower) kernel path" % head_dim)
        self.batch_first = batch_first
        self.im2col_step = img2col_step
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.num_levels = num_levels
        self.num_points = num_points
        self.sampling_offsets = nn.Linear(embed_dim, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dim, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dim, embed_dim)
        self.output_proj = nn.Linear(embed_dim, embed_dim)
        self.init_weights()
        self.register_load_state_dict_post_hook(MultiScaleDeformableAttention.refresh_fused_projection)

    def _reset_parameters(self):
        return self.init_weights()

    def init_weights(self):
        """Deterministic init of the reference (ms_deform_attn.py:194-217): zero offset weights,
        bias = M unit directions (max-norm 1) scaled by the point index, uniform attention."""
        with torch.no_grad():
            self.sampling_offsets.weight.zero_()
            theta = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
            direction = torch.stack([theta.cos(), theta.sin()], -1)
            direction = direction / direction.abs().max(-1, keepdim=True)[0]
            grid = direction.view(self.num_heads, 1, 1, 2).repeat(1, self.num_levels, self.num_points, 1)
            grid = grid * torch.arange(1, self.num_points + 1, dtype=torch.float32).view(1, 1, -1, 1)
            self.sampling_offsets.bias.copy_(grid.reshape(-1))
            self.attention_weights.weight.zero_()
            self.attention_weights.bias.zero_()
            nn.init.xavier_uniform_(self.value_proj.weight)
            self.value_proj.bias.zero_()
            nn.init.xavier_uniform_(self.output_proj.weight)
            self.output_proj.bias.zero_()

    def freeze_sampling_offsets(self):
        print("Freeze sampling offsets")
        self.sampling_offsets.weight.requires_grad = False
        self.sampling_offsets.bias.requires_grad = False

    def freeze_attention_weights(self):
        print("Freeze attention weights")
        self.attention_weights.weight.requires_grad = False
        self.attention_weights.bias.requires_grad = False

    def refresh_fused_projection(self, *unused):
        """Bring the concatenated copy of the two query projections up to date (in place).  Runs after every
        ``load_state_dict`` (post-hook registered in ``__init__``); ``GraphedTransformer`` also calls it when a parameter
        version changed, because a graph replay skips the Python code that would notice."""
        if getattr(self, "_fused_qp", None) is not None:
            self._fused_query_projection()
        if "_bf16x3_split" in self.__dict__:   # the bf16 planes of the projections (gemm_bf16x3.py) follow too
            from . import gemm_bf16x3 as g3
            fq = self._fused_query_projection()
            g3.refresh(self, {"value": self.value_proj.weight, "output": self.output_proj.weight,
                              **({"query": fq[0]} if fq is not None else {})})
        # (the planes _MultiValueProjections keeps on the value projection's own Parameter: the decoder's batched projections
        #  of the memory are replayed from a graph as well)
        w = self.value_proj.weight
        if "_bf16x3_split" in w.__dict__:
            from . import gemm_bf16x3 as g3
            g3.refresh(w, {"self": w})

    fuse_query_projections = True   # class-level switch (tests compare both ways)
    fuse_sampling_plan = True       # ... softmax + sampling locations in one native launch each way (needs the fused projection)

    def _fused_query_projection(self):
        """``sampling_offsets`` and ``attention_weights`` read the same query: while both are frozen (every ZiRa task)
        their weights are kept concatenated -- rebuilt when either tensor changes (load_state_dict, .to())."""
        so, aw = self.sampling_offsets, self.attention_weights
        if (not self.fuse_query_projections or so.weight.requires_grad or aw.weight.requires_grad
                or so.bias is None or aw.bias is None or so.bias.requires_grad or aw.bias.requires_grad
                or not so.weight.is_cuda):
            return None
        key = (so.weight.data_ptr(), so.weight._version, aw.weight.data_ptr(), aw.weight._version,
               so.bias.data_ptr(), so.bias._version, aw.bias.data_ptr(), aw.bias._version, so.weight.dtype)
        cached = getattr(self, "_fused_qp", None)
        if cached is None or cached[0] != key:
            with torch.no_grad():
                n1 = so.weight.shape[0]
                if (cached is not None and cached[1].device == so.weight.device and cached[1].dtype == so.weight.dtype
                        and cached[1].shape[0] == n1 + aw.weight.shape[0] and cached[1].shape[1] == so.weight.shape[1]):
                    # refreshed IN PLACE: captured hipGraphs (GraphedTransformer) keep reading these two buffers, and
                    # a replay never re-runs this Python code
                    w, b = cached[1], cached[2]
                    w[:n1].copy_(so.weight)
                    w[n1:].copy_(aw.weight)
                    b[:n1].copy_(so.bias)
                    b[n1:].copy_(aw.bias)
                else:
                    w = torch.cat([so.weight, aw.weight], 0).contiguous()
                    b = torch.cat([so.bias, aw.bias], 0).contiguous()
            cached = self._fused_qp = (key, w, b)
        return cached[1], cached[2]

    fuse_shared_source = True       # value == query source (encoder self-attention): both projections one autograd node

    def project(self, query, value, key_padding_mask, reference_points, spatial_shapes, shared_source=None,
                value_projected=None):
        """Everything of ``forward`` up to the native call (reference :286-325), batch-first.
        Returns (value[B,S,M,D], sampling_locations[B,Q,M,L,P,2], attention_weights[B,Q,M,L,P]).
        ``shared_source`` = (src, pos) when ``value is src`` and ``query = src + pos`` (``query`` may then be None);
        ``value_projected``: ``value_proj(value)`` with the padding zeroed, made elsewhere (``multi_value_projections``)."""
        M, L, P = self.num_heads, self.num_levels, self.num_points
        fused = self._fused_query_projection()
        oa = None
        if (shared_source is not None and fused is not None and self.fuse_shared_source and value_projected is None
                and _frozen_fp32_linear(self.value_proj, shared_source[0])):
            src, pos = shared_source
            value, oa = _SharedSourceProjections.apply(src, pos, self.value_proj.weight, self.value_proj.bias, fused[0], fused[1])
            bs, num_query = src.shape[0], src.shape[1]
            num_value = num_query
        else:
            if query is None:
                query = shared_source[0] if shared_source[1] is None else shared_source[0] + shared_source[1]
            bs, num_query, _ = query.shape
            num_value = value.shape[1]
            value = self.value_proj(value) if value_projected is None else value_projected
        if key_padding_mask is not None and value_projected is None:
            value = value.masked_fill(key_padding_mask[..., None], float(0))
        value = value.view(bs, num_value, M, -1)
        if fused is not None:   # one GEMM for the two projections of the query (both frozen: one dgrad GEMM, no add)
            if oa is None:
                oa = F.linear(query, fused[0], fused[1])
            if self.fuse_sampling_plan and _sampling_plan_ok(oa, reference_points, spatial_shapes, L, P):
                loc, attn = _SamplingPlan.apply(oa, reference_points, spatial_shapes, M, L, P)
                return value, loc, attn
            offsets, attn = oa.split([M * L * P * 2, M * L * P], dim=-1)   # (split: its backward is one cat)
            offsets = offsets.reshape(bs, num_query, M, L, P, 2)
            attn = attn.reshape(bs, num_query, M, L * P)
        else:
            offsets = self.sampling_offsets(query).view(bs, num_query, M, L, P, 2)
            attn = self.attention_weights(query).view(bs, num_query, M, L * P)
        attn = attn.softmax(-1).view(bs, num_query, M, L, P)
        loc = sampling_locations_from_reference_points(reference_points, offsets, spatial_shapes, P)
        return value, loc, attn

    def forward(self, query: torch.Tensor, key: Optional[torch.Tensor] = None,
                value: Optional[torch.Tensor] = None, query_pos: Optional[torch.Tensor] = None,
                key_padding_mask: Optional[torch.Tensor] = None,
                reference_points: Optional[torch.Tensor] = None,
                spatial_shapes: Optional[torch.Tensor] = None,
                level_start_index: Optional[torch.Tensor] = None,
                value_projected: Optional[torch.Tensor] = None, **kwargs) -> torch.Tensor:
        if value is None:
            value = query
        # value and query from ONE tensor (the encoder's self-attention: query = src + pos, value = src): both projections
        # become one autograd node (see _SharedSourceProjections)
        shared = (query, query_pos) if (value is query and self.batch_first and value_projected is None) else None
        if shared is not None:
            query = None
        elif query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query = query.permute(1, 0, 2)
            value = value.permute(1, 0, 2)
            if value_projected is not None and value_projected.shape[0] != value.shape[0]:
                value_projected = value_projected.permute(1, 0, 2)
        _check_levels_cover_value(spatial_shapes, value.shape[1], level_start_index)

        value, loc, attn = self.project(query, value, key_padding_mask, reference_points,
                                        spatial_shapes, shared_source=shared, value_projected=value_projected)
        half = value.dtype in (torch.float16, torch.bfloat16)
        out_dtype = value.dtype
        if half:  # the native op is fp32/fp64 (reference :326-344 upcasts fp16 the same way)
            value, loc, attn = value.float(), loc.float(), attn.float()
        if value.is_cuda:
            output = MultiScaleDeformableAttnFunction.apply(
                value.contiguous(), spatial_shapes, level_start_index, loc.contiguous(),
                attn.contiguous(), self.im2col_step)
        else:  # CPU tensors: the pure-PyTorch path, as in the reference module (:345-348)
            output = multi_scale_deformable_attn_pytorch(value, spatial_shapes, loc, attn)
        if half:
            output = output.to(out_dtype)
        output = self.output_proj(output)
        if not self.batch_first:
            output = output.permute(1, 0, 2)
        return output
