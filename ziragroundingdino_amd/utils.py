"""Small building blocks of the cross-modal transformer, with the reference's names.

Mirrors groundingdino/models/GroundingDINO/utils.py and util/misc.py (only what the hot path
uses): ``MLP`` (:171), ``ContrastiveEmbed`` (:234-269), ``recover_to_cls_logits`` (:312-320),
``gen_sineembed_for_position`` (:204-231), ``get_sine_pos_embed`` (:24-53),
``gen_encoder_output_proposals`` (:56-116), ``inverse_sigmoid`` (util/misc.py:704-708),
``NestedTensor`` / ``nested_tensor_from_tensor_list`` (util/misc.py:440-499).

Differences that do not change results: ``recover_to_cls_logits`` is one masked max per image
instead of a Python loop over categories (the reference loops B x n_cat with boolean indexing,
each iteration a device sync).
"""
import copy
import math
from typing import List, Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn


def _get_clones(module, N, layer_share=False):
    if layer_share:
        return nn.ModuleList([module for _ in range(N)])
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


def _get_activation_fn(activation, d_model=256, batch_dim=0):
    table = {"relu": F.relu, "gelu": F.gelu, "glu": F.glu, "selu": F.selu}
    if activation in table:
        return table[activation]
    if activation == "prelu":
        return nn.PReLU()
    raise RuntimeError(f"activation should be relu/gelu, not {activation}.")


def _native_elementwise_ok(x: Tensor) -> bool:
    """fp32 GPU tensors that take no part in autograd: the one-launch HIP form of the sine embedding applies
    (csrc/refpoints.hip)."""
    return (x.is_cuda and x.dtype == torch.float32 and not (x.requires_grad and torch.is_grad_enabled())
            and x.numel() > 0 and NATIVE_REFPOINT_OPS)


NATIVE_REFPOINT_OPS = True   # developer switch: False = the PyTorch op chains everywhere


def inverse_sigmoid(x, eps=1e-3):
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


class _BoxHead(torch.autograd.Function):
    """sigmoid(delta + inverse_sigmoid(ref)) as one node: csrc/refpoints.hip, one launch forward and one backward instead of
    8 + ~20 ATen kernels (the box head of groundingdino_dual_zero_rep_branch.py:563-569 over all decoder layers at once)."""

    @staticmethod
    def forward(ctx, delta, ref, eps):
        from . import _lib
        delta, ref = delta.contiguous(), ref.contiguous()
        out = torch.empty_like(delta)
        with torch.cuda.device(delta.device):
            rc = _lib.load().zira_box_head_fwd_f32(delta.data_ptr(), ref.data_ptr(), delta.numel(), float(eps), out.data_ptr(),
                                                   torch.cuda.current_stream(delta.device).cuda_stream)
        if rc != 0:
            raise RuntimeError("zira_box_head_fwd_f32 failed: hipError %d" % rc)
        ctx.save_for_backward(out, ref)
        ctx.eps = float(eps)
        return out

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        out, ref = ctx.saved_tensors
        g = g.contiguous()
        g_delta = torch.empty_like(out) if ctx.needs_input_grad[0] else None
        g_ref = torch.empty_like(out) if ctx.needs_input_grad[1] else None
        if g_delta is not None or g_ref is not None:
            with torch.cuda.device(out.device):
                rc = _lib.load().zira_box_head_bwd_f32(g.data_ptr(), out.data_ptr(), ref.data_ptr(), out.numel(), ctx.eps,
                                                       0 if g_delta is None else g_delta.data_ptr(),
                                                       0 if g_ref is None else g_ref.data_ptr(),
                                                       torch.cuda.current_stream(out.device).cuda_stream)
            if rc != 0:
                raise RuntimeError("zira_box_head_bwd_f32 failed: hipError %d" % rc)
        return g_delta, g_ref, None


def box_head(delta, ref, eps=1e-3):
    """``(delta + inverse_sigmoid(ref, eps)).sigmoid()``; fp32 GPU tensors of one shape go through one native node."""
    if (delta.is_cuda and delta.dtype == torch.float32 and ref.dtype == torch.float32 and delta.shape == ref.shape
            and ref.device == delta.device and not torch.is_autocast_enabled("cuda")):
        return _BoxHead.apply(delta, ref, eps)
    return (delta + inverse_sigmoid(ref, eps)).sigmoid()


class MLP(nn.Module):
    """num_layers Linear layers with ReLU in between (state-dict keys ``layers.{i}.*``)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [output_dim]
        self.layers = nn.ModuleList(nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = layer(x)
            if i < self.num_layers - 1:
                x = F.relu(x)
        return x

    def refresh_fused_projection(self, *unused):
        """Transposed weight copies kept by the native decoder path (decoder_layer.MLPWeights) follow the parameters in
        place (GraphedTransformer calls this when a parameter changed: a replayed graph re-runs no Python)."""
        w = getattr(self, "_native_weights", None)
        if w is not None:
            w.refresh(self)


_SINE_CONSTS = {}


def _interleaved_sincos(arg: Tensor) -> Tensor:
    """[..., n] -> [..., n] with sin on even and cos on odd channels (pairs share a frequency)."""
    return torch.stack((arg[..., 0::2].sin(), arg[..., 1::2].cos()), dim=-1).flatten(-2)


def _dim_t(n: int, temperature: float, device) -> Tensor:
    i = torch.arange(n, dtype=torch.float32, device=device)
    return temperature ** (2 * torch.div(i, 2, rounding_mode="floor") / n)


def get_sine_pos_embed(pos_tensor: Tensor, num_pos_feats: int = 128, temperature: int = 10000,
                       exchange_xy: bool = True) -> Tensor:
    """[..., n] positions -> [..., n*num_pos_feats] sine embedding (reference utils.py:24-53)."""
    dim_t = _dim_t(num_pos_feats, temperature, pos_tensor.device)
    parts = [_interleaved_sincos(x * (2 * math.pi) / dim_t)
             for x in pos_tensor.split([1] * pos_tensor.shape[-1], dim=-1)]
    if exchange_xy:
        parts[0], parts[1] = parts[1], parts[0]
    return torch.cat(parts, dim=-1)


def gen_sineembed_for_position(pos_tensor: Tensor) -> Tensor:
    """[nq, bs, 2|4] (x, y[, w, h]) in [0,1] -> [nq, bs, 256|512], ordered (y, x[, w, h]);
    128 features each, temperature 10000 (reference utils.py:204-231)."""
    n = pos_tensor.size(-1)
    if n not in (2, 4):
        raise ValueError("Unknown pos_tensor shape(-1):{}".format(n))
    # all coordinates at once (same arithmetic per element as the reference's per-coordinate loop,
    # 6 kernels instead of 30): reorder (x, y, ...) -> (y, x, ...), scale, divide, sin / cos
    key = (str(pos_tensor.device), n)
    consts = _SINE_CONSTS.get(key)
    if consts is None:
        order = torch.tensor([1, 0, 2, 3][:n], device=pos_tensor.device)
        consts = _SINE_CONSTS[key] = (order, _dim_t(128, 10000, pos_tensor.device))
    order, dim_t = consts
    if _native_elementwise_ok(pos_tensor):
        from . import _lib

        pc = pos_tensor.contiguous()
        out = torch.empty(pc.shape[:-1] + (n * 128,), dtype=torch.float32, device=pc.device)
        with torch.cuda.device(pc.device):
            rc = _lib.load().zira_sine_embed_f32(pc.data_ptr(), dim_t.data_ptr(), pc.numel() // n, n, 128, 2 * math.pi,
                                                 out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("zira_sine_embed_f32 failed: hipError %d" % rc)
        return out
    arg = pos_tensor.index_select(-1, order)[..., None] * (2 * math.pi) / dim_t      # [nq, bs, n, 128]
    return _interleaved_sincos(arg).flatten(2)


def gen_encoder_output_proposals(memory: Tensor, memory_padding_mask: Tensor, spatial_shapes,
                                 learnedwh=None):
    """Two-stage proposals: one box per pixel, centre = pixel centre / valid size, side
    0.05 * 2^level; returned un-sigmoided, +inf where padded or outside (0.01, 0.99)
    (reference utils.py:56-116).  ``spatial_shapes`` may be a tensor or a list of (H, W)."""
    N, S, C = memory.shape
    shapes = [(int(h), int(w)) for h, w in (spatial_shapes.tolist() if torch.is_tensor(spatial_shapes)
                                            else spatial_shapes)]
    if learnedwh is None and memory.is_cuda:
        from . import geometry
        from .transformer import Switches
        if Switches.native_geometry and geometry.supported(memory_padding_mask, shapes):
            # two launches and ATen's log instead of ~60 launch-bound kernels (csrc/refpoints.hip), bit-identical
            output_proposals, drop = geometry.encoder_proposals(memory_padding_mask, shapes)
            return memory.masked_fill(drop.unsqueeze(-1), 0.0), output_proposals
    proposals = []
    cur = 0
    for lvl, (H, W) in enumerate(shapes):
        m = memory_padding_mask[:, cur:cur + H * W].view(N, H, W)
        valid_H = (~m[:, :, 0]).sum(1)
        valid_W = (~m[:, 0, :]).sum(1)
        gy, gx = torch.meshgrid(
            torch.linspace(0, H - 1, H, dtype=torch.float32, device=memory.device),
            torch.linspace(0, W - 1, W, dtype=torch.float32, device=memory.device), indexing="ij")
        grid = torch.stack([gx, gy], -1)                                   # H, W, 2 (x, y)
        scale = torch.stack([valid_W, valid_H], 1).view(N, 1, 1, 2)
        grid = (grid.unsqueeze(0).expand(N, -1, -1, -1) + 0.5) / scale
        if learnedwh is not None:
            wh = torch.ones_like(grid) * learnedwh.sigmoid() * (2.0 ** lvl)
        else:
            wh = torch.ones_like(grid) * 0.05 * (2.0 ** lvl)
        proposals.append(torch.cat((grid, wh), -1).view(N, -1, 4))
        cur += H * W
    output_proposals = torch.cat(proposals, 1)
    valid = ((output_proposals > 0.01) & (output_proposals < 0.99)).all(-1, keepdim=True)
    output_proposals = torch.log(output_proposals / (1 - output_proposals))
    output_proposals = output_proposals.masked_fill(memory_padding_mask.unsqueeze(-1), float("inf"))
    output_proposals = output_proposals.masked_fill(~valid, float("inf"))
    # (one fill with the union of the two masks instead of the reference's two passes over the 45 MB tokens -- and two more
    #  in the backward; the same zeros)
    # (torch.where was measured here and lost: its backward runs two broadcasting kernels of 75 us each)
    output_memory = memory.masked_fill(memory_padding_mask.unsqueeze(-1) | ~valid, 0.0)
    return output_memory, output_proposals


class ContrastiveEmbed(nn.Module):
    """Parameter-free classifier: logits[b, q, t] = <x[b, q], text[b, t]>, -inf on padded
    tokens and padded out to ``max_text_len`` columns (reference utils.py:234-269)."""

    def __init__(self, max_text_len=256):
        super().__init__()
        self.max_text_len = max_text_len

    def forward(self, x, text_dict):
        assert isinstance(text_dict, dict)
        y = text_dict["encoded_text"]
        text_token_mask = text_dict["text_token_mask"]
        res = x @ y.transpose(-1, -2)
        res = res.masked_fill(~text_token_mask[:, None, :], float("-inf"))
        pad = self.max_text_len - res.shape[-1]
        if pad > 0:
            res = F.pad(res, (0, pad), value=float("-inf"))
        return res


_mask_pack_cache = {}


def _packed_category_masks(masks: List[Tensor], device):
    """(mask uint8 [B, Cmax, Tmax], n_cat int32 [B], n_tok int32 [B], Cmax, Tmax) on ``device`` for the native
    kernel; remembered per list object (the model caches the list per caption batch)."""
    key = (id(masks), str(device))
    hit = _mask_pack_cache.get(key)
    if hit is not None and hit[0] is masks:
        return hit[1]
    cmax = max([m.shape[0] for m in masks] + [0])
    tmax = max([m.shape[1] for m in masks] + [0])
    packed = torch.zeros((len(masks), cmax, tmax), dtype=torch.uint8)
    for b, m in enumerate(masks):
        packed[b, :m.shape[0], :m.shape[1]] = m.to("cpu", torch.uint8)
    val = (packed.to(device), torch.tensor([m.shape[0] for m in masks], dtype=torch.int32).to(device),
           torch.tensor([m.shape[1] for m in masks], dtype=torch.int32).to(device), cmax, tmax)
    if len(_mask_pack_cache) > 64:
        _mask_pack_cache.clear()
    _mask_pack_cache[key] = (masks, val)
    return val


class _CategoryLogits(torch.autograd.Function):
    """recover_to_cls_logits on the GPU through the C ABI (csrc/catlogits.hip): one kernel each way."""

    @staticmethod
    def forward(ctx, logits, packed, n_cat, n_tok, cmax, tmax, for_fill):
        from . import _lib

        lib = _lib.load()
        logits = logits.contiguous()
        B, Q, T = logits.shape[-3:]
        rows = logits.numel() // T
        out = torch.empty_like(logits)
        arg = torch.empty((rows, max(cmax, 1)), dtype=torch.int32, device=logits.device)
        with torch.cuda.device(logits.device):
            rc = lib.zira_cat_logits_fwd_f32(logits.data_ptr(), packed.data_ptr(), n_cat.data_ptr(), n_tok.data_ptr(), rows,
                                             B, Q, T, cmax, tmax, float(for_fill), out.data_ptr(), arg.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("zira_cat_logits_fwd_f32 failed with code %d" % rc)
        ctx.save_for_backward(arg, n_cat)
        ctx.dims = (rows, B, Q, T, cmax)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        from . import _lib

        lib = _lib.load()
        arg, n_cat = ctx.saved_tensors
        rows, B, Q, T, cmax = ctx.dims
        grad_out = grad_out.contiguous()
        grad = torch.empty_like(grad_out)
        with torch.cuda.device(grad_out.device):
            rc = lib.zira_cat_logits_bwd_f32(grad_out.data_ptr(), arg.data_ptr(), n_cat.data_ptr(), rows, B, Q, T, cmax,
                                             grad.data_ptr(), torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("zira_cat_logits_bwd_f32 failed with code %d" % rc)
        return grad, None, None, None, None, None, None


def recover_to_cls_logits(logits: Tensor, cate_to_token_mask_list: List[Tensor],
                          for_fill=float("-inf")) -> Tensor:
    """token logits -> category logits: new[b, q, c] = max over the tokens of category c,
    ``for_fill`` elsewhere; same shape as ``logits`` (reference utils.py:312-320)."""
    assert logits.shape[-3] == len(cate_to_token_mask_list)    # [..., B, Q, T]: leading dims = stacked sets
    if (logits.is_cuda and logits.dtype == torch.float32
            and all(m.shape[1] <= logits.shape[-1] and m.shape[0] <= logits.shape[-1] for m in cate_to_token_mask_list)):
        packed, n_cat, n_tok, cmax, tmax = _packed_category_masks(cate_to_token_mask_list, logits.device)
        return _CategoryLogits.apply(logits, packed, n_cat, n_tok, cmax, tmax, for_fill)
    new_logits = torch.full(logits.shape, for_fill, device=logits.device, dtype=logits.dtype)
    for bid, mask in enumerate(cate_to_token_mask_list):          # mask: [n_cat, n_token] bool
        n_cat, n_tok = mask.shape
        if n_cat == 0:
            continue
        tok = logits[..., bid, :, :n_tok]                           # [..., Q, n_tok]
        per_cat = tok[..., None, :].masked_fill(~mask, float("-inf")).max(dim=-1)[0]
        # A category without tokens (the mask generator emits one between the last "." and [SEP]
        # whenever the caption is shorter than the batch's longest, bertwarper.py:262-266) makes
        # the reference's max() over an empty selection raise; here it reads as "no such category".
        per_cat = per_cat.masked_fill(torch.isneginf(per_cat), for_fill)
        new_logits[..., bid, :, :n_cat] = per_cat
    return new_logits


class NestedTensor(object):
    """(tensors [B,C,H,W], mask [B,H,W] True = padding) pair (reference util/misc.py:440-470)."""

    def __init__(self, tensors, mask: Optional[Tensor]):
        self.tensors = tensors
        self.mask = mask
        self.no_padding = False  # True: all images fill the batch tensor (mask is all False)

    def to(self, device):
        mask = self.mask.to(device) if self.mask is not None else None
        return NestedTensor(self.tensors.to(device), mask)

    def decompose(self):
        return self.tensors, self.mask

    @property
    def device(self):
        return self.tensors.device


def nested_tensor_from_tensor_list(tensor_list) -> NestedTensor:
    """Pad a list of [C,Hi,Wi] images to the common max size (reference util/misc.py:474-499)."""
    if hasattr(tensor_list, "tensor") and hasattr(tensor_list, "image_sizes"):  # ImageList-like
        sizes = tensor_list.image_sizes
        batch = tensor_list.tensor
        mask = torch.ones(batch.shape[0], batch.shape[2], batch.shape[3], dtype=torch.bool,
                          device=batch.device)
        for m, (h, w) in zip(mask, sizes):
            m[:h, :w] = False
        out = NestedTensor(batch, mask)
        out.no_padding = all((h, w) == tuple(batch.shape[2:]) for h, w in sizes)  # known on the host
        return out
    if tensor_list[0].ndim != 3:
        raise ValueError("not supported")
    c = tensor_list[0].shape[0]
    h = max(img.shape[1] for img in tensor_list)
    w = max(img.shape[2] for img in tensor_list)
    tensor = torch.zeros((len(tensor_list), c, h, w), dtype=tensor_list[0].dtype,
                         device=tensor_list[0].device)
    mask = torch.ones((len(tensor_list), h, w), dtype=torch.bool, device=tensor_list[0].device)
    for img, pad_img, m in zip(tensor_list, tensor, mask):
        pad_img[:, :img.shape[1], :img.shape[2]].copy_(img)
        m[:img.shape[1], :img.shape[2]] = False
    out = NestedTensor(tensor, mask)
    out.no_padding = all(tuple(img.shape[1:]) == (h, w) for img in tensor_list)  # known on the host
    return out
