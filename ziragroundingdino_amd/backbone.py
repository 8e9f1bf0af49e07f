"""Frozen image backbone: Swin Transformer + sine position encoding.

Out of kernel scope: the backbone is frozen and has no backward in
ZiRa, its cost is dense GEMMs / window attention that stock PyTorch-ROCm already sends to MFMA
(hipBLASLt, fused SDPA).  It exists here because the end-to-end training step the bench times
starts at the pixels.  Architecture, feature shapes and parameter names follow the reference
(groundingdino/models/GroundingDINO/backbone/swin_transformer.py, position_encoding.py:78-134,
backbone.py:146-221) so that its checkpoints load: ``patch_embed.proj``, ``layers.N.blocks.K.
{norm1, attn.{relative_position_bias_table, qkv, proj}, norm2, mlp.{fc1, fc2}}``,
``layers.N.downsample.{reduction, norm}``, ``norm{1,2,3}``.
"""
import math
from typing import List

import torch
import torch.nn.functional as F
from torch import nn

from .dense import LayerNorm, conv_module_as_gemm
from .transformer import DropPath
from .utils import NestedTensor

SWIN_VARIANTS = {  # reference swin_transformer.py:772-788
    "swin_T_224_1k": dict(embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=7),
    "swin_B_224_22k": dict(embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=7),
    "swin_B_384_22k": dict(embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=12),
    "swin_L_224_22k": dict(embed_dim=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48], window_size=7),
    "swin_L_384_22k": dict(embed_dim=192, depths=[2, 2, 18, 2], num_heads=[6, 12, 24, 48], window_size=12),
}


FUSED_EPILOGUES = True    # developer switch (scripts/ab_step.py swin_epilogues=0): GELU and the residuals as kernels of their own
MIN_ROWS = 8192           # (at the last stage's 2100 rows the library's kernels are 1.2-1.8 x faster, scripts/gemm_bf16x3_swin.py)


def _frozen_ok(lin, x):
    from . import gemm_bf16x3 as g3
    return (g3.enabled() and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and lin.bias is not None
            and not lin.weight.requires_grad and not torch.is_autocast_enabled("cuda"))


def _frozen_linear(owner, name, lin, x, epilogue=None, residual=None, scale=None):
    """``lin(x)`` for a frozen Linear of the backbone: under ``transformer.Switches.gemm_arith`` = "bf16x3" / "f16x2" the
    fp32-accurate split product on the matrix cores (gemm_bf16x3.py; the weight's planes are cached on ``owner`` and follow the
    parameter in place) from 8192 rows, else the library.

    With "f16x2" the tiled kernel also takes the widths that are multiples of 32 only (Swin-T's 96, 192, 288, 576) and what
    follows the product in a block: ``epilogue="gelu"`` (the MLP's activation) or ``residual`` [.., N] (+ ``scale`` [B, 1, 1],
    the stochastic-depth factor per image): ``residual + scale * lin(x)`` -- one launch instead of GEMM + GELU / GEMM + addcmul,
    and no round trip of the 4x-wide hidden activation.  Returns None when an epilogue was asked for and the fused kernel does
    not take the call (the caller composes it from ATen ops then)."""
    from . import gemm_bf16x3 as g3
    fused = epilogue is not None or residual is not None
    if _frozen_ok(lin, x):
        x2 = x.reshape(-1, x.shape[-1])
        N, K = lin.weight.shape
        if x2.shape[0] >= MIN_ROWS:
            if fused:
                if FUSED_EPILOGUES and g3.supported_f16x2(x2.contiguous(), N, K):
                    if residual is None:
                        out = g3.linear_tiled_f16x2(owner, name, x2.contiguous(), lin.weight, lin.bias, g3.EPI_BIAS_GELU)
                    else:
                        r2 = residual.reshape(-1, N).contiguous()
                        rs = None if scale is None else scale.reshape(-1).contiguous()
                        out = g3.linear_tiled_f16x2(owner, name, x2.contiguous(), lin.weight, lin.bias, g3.EPI_BIAS_RES, residual=r2,
                                                    row_scale=rs, rows_per_scale=0 if rs is None else x2.shape[0] // rs.numel())
                    return out.view(*x.shape[:-1], -1)
                return None
            if g3.supported(x2.contiguous(), N, K):
                return g3.linear(owner, name, x2.contiguous(), lin.weight, lin.bias).view(*x.shape[:-1], -1)
            if g3.supported_f16x2(x2.contiguous(), N, K):
                return g3.linear_tiled_f16x2(owner, name, x2.contiguous(), lin.weight, lin.bias).view(*x.shape[:-1], -1)
    return None if fused else lin(x)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, in_features)

    def forward(self, x, residual=None, scale=None):
        """fc2(gelu(fc1(x))); with ``residual``: residual + scale * that (``scale`` [B, 1, 1] or None), the block's second half."""
        h = _frozen_linear(self, "fc1", self.fc1, x, epilogue="gelu")
        if h is None:
            h = self.act(_frozen_linear(self, "fc1", self.fc1, x))
        if residual is None:
            return _frozen_linear(self, "fc2", self.fc2, h)
        out = _frozen_linear(self, "fc2", self.fc2, h, residual=residual, scale=scale)
        if out is None:
            y = _frozen_linear(self, "fc2", self.fc2, h)
            out = residual + y if scale is None else torch.addcmul(residual, y, scale)
        return out


def window_partition(x, ws):
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)


def window_reverse(windows, ws, H, W):
    B = windows.shape[0] // ((H // ws) * (W // ws))
    x = windows.view(B, H // ws, W // ws, ws, ws, -1)
    return x.permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads):
        super().__init__()
        self.dim, self.ws, self.num_heads = dim, window_size, num_heads
        self.relative_position_bias_table = nn.Parameter(
            torch.zeros((2 * window_size - 1) * (2 * window_size - 1), num_heads))
        coords = torch.stack(torch.meshgrid(torch.arange(window_size), torch.arange(window_size), indexing="ij"))
        coords = coords.flatten(1)
        rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += window_size - 1
        rel[:, :, 1] += window_size - 1
        rel[:, :, 0] *= 2 * window_size - 1
        self.register_buffer("relative_position_index", rel.sum(-1))
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)

    def _bias_transposed(self):
        """[heads, N, N] relative position bias with the key index first (bias_t[h, j, i]), as the native kernel
        reads it; rebuilt when the (frozen) table changes."""
        t = self.relative_position_bias_table
        key = (t._version, t.device, t.data_ptr())
        if getattr(self, "_bias_t_key", None) != key:
            N = self.ws * self.ws
            with torch.no_grad():
                self._bias_t = t[self.relative_position_index.view(-1)].view(N, N, -1).permute(2, 1, 0).contiguous().float()
            self._bias_t_key = key
        return self._bias_t

    def forward_native(self, xn, H, W, shift, residual=None, scale=None):
        """``proj(window attention(qkv(xn)))`` for the normalised token map ``xn [B, H*W, C]`` through the C ABI
        (csrc/winattn.hip): pad, shift, window partition / reverse and crop are index arithmetic inside the kernel.
        Forward only (the backbone is frozen)."""
        from . import _lib

        lib = _lib.load()
        B, L, C = xn.shape
        qkv = _frozen_linear(self, "qkv", self.qkv, xn).contiguous()   # [B, H, W, 3, heads, 32]; bfloat16 under bf16 autocast
        hd = C // self.num_heads
        if qkv.dtype == torch.bfloat16:                    # (12 x 12 windows only: SwinTransformerBlock.forward checks)
            fn, name = lib.zira_window_attn_bf16, "zira_window_attn_bf16"
        elif qkv.dtype == torch.float32:
            fn, name = lib.zira_window_attn_f32, "zira_window_attn_f32"
        else:
            raise RuntimeError("window attention: qkv must be float32 or bfloat16, got %s" % qkv.dtype)
        out = torch.empty((B, L, C), dtype=qkv.dtype, device=xn.device)
        with torch.cuda.device(xn.device):
            rc = fn(qkv.data_ptr(), self.qkv.bias.data_ptr(), self._bias_transposed().data_ptr(),
                    B, H, W, self.num_heads, hd, self.ws, shift, float(hd) ** -0.5, out.data_ptr(),
                    torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("%s failed with code %d" % (name, rc))
        if residual is None:
            return _frozen_linear(self, "proj", self.proj, out)
        y = _frozen_linear(self, "proj", self.proj, out, residual=residual, scale=scale)   # the block's first residual in the epilogue
        if y is None:
            y = _frozen_linear(self, "proj", self.proj, out)
            y = residual + y if scale is None else torch.addcmul(residual, y, scale)
        return y

    def forward(self, x, mask=None):
        Bw, N, C = x.shape
        qkv = self.qkv(x).reshape(Bw, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        bias = self.relative_position_bias_table[self.relative_position_index.view(-1)]
        bias = bias.view(N, N, -1).permute(2, 0, 1).unsqueeze(0)                # 1, heads, N, N
        if mask is not None:                                                    # nW, N, N
            nW = mask.shape[0]
            bias = (bias + mask.unsqueeze(1)).repeat(Bw // nW, 1, 1, 1)         # Bw, heads, N, N
        x = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2], attn_mask=bias.to(x.dtype))
        return self.proj(x.transpose(1, 2).reshape(Bw, N, C))


class SwinTransformerBlock(nn.Module):
    native_max_tokens = 64
    native_attention = True   # no-grad GPU forwards: window attention through the C ABI (csrc/winattn.hip) -- fp32 with
                              # windows of <= 64 tokens (7x7: one query row per lane) and, fp32 or under bf16 autocast,
                              # the 12x12 windows of the 384-pixel Swin-B / L variants (144 tokens: the MFMA kernel)

    def _native_ok(self, x, C):
        ws = self.window_size
        if not (self.native_attention and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()
                and C // self.attn.num_heads == 32 and self.attn.qkv.bias is not None
                and self.attn.qkv.weight.dtype == torch.float32 and self.attn.qkv.bias.dtype == torch.float32):
            return False
        if torch.is_autocast_enabled():
            return ws == 12 and torch.get_autocast_dtype("cuda") == torch.bfloat16
        return ws * ws <= self.native_max_tokens or ws == 12

    def __init__(self, dim, num_heads, window_size, shift_size, mlp_ratio, drop_path):
        super().__init__()
        self.window_size, self.shift_size = window_size, shift_size
        self.norm1 = LayerNorm(dim)
        self.attn = WindowAttention(dim, window_size, num_heads)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def _residual(self, x, branch, scale):
        """x + drop_path(branch); ``scale`` [B, 1, 1] is a pre-drawn keep / keep_prob factor (SwinTransformer.forward
        draws the factors of all blocks with four launches instead of three per use)."""
        if scale is None:
            return x + self.drop_path(branch)
        return torch.addcmul(x, branch, scale)

    def forward(self, x, H, W, mask_matrix, dp=None):
        B, L, C = x.shape
        ws = self.window_size
        shortcut = x
        dp0, dp1 = (dp[0], dp[1]) if dp is not None else (None, None)
        if self._native_ok(x, C):
            plain = isinstance(self.drop_path, nn.Identity) or not self.training
            if dp0 is not None or plain:      # (a scale drawn by the caller, or none to draw: the residuals ride in the GEMMs)
                x = self.attn.forward_native(self.norm1(x), H, W, self.shift_size, residual=shortcut, scale=dp0)
                return self.mlp(self.norm2(x), residual=x, scale=dp1)
            x = self._residual(shortcut, self.attn.forward_native(self.norm1(x), H, W, self.shift_size), dp0)
            return self._residual(x, self.mlp(self.norm2(x)), dp1)
        x = self.norm1(x).view(B, H, W, C)
        pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
        x = F.pad(x, (0, 0, 0, pad_r, 0, pad_b))
        Hp, Wp = H + pad_b, W + pad_r
        if self.shift_size > 0:
            x = torch.roll(x, shifts=(-self.shift_size, -self.shift_size), dims=(1, 2))
            attn_mask = mask_matrix() if callable(mask_matrix) else mask_matrix
        else:
            attn_mask = None
        w = self.attn(window_partition(x, ws), mask=attn_mask)
        x = window_reverse(w, ws, Hp, Wp)
        if self.shift_size > 0:
            x = torch.roll(x, shifts=(self.shift_size, self.shift_size), dims=(1, 2))
        x = x[:, :H, :W, :].reshape(B, H * W, C)
        x = self._residual(shortcut, x, dp0)
        return self._residual(x, self.mlp(self.norm2(x)), dp1)


class PatchMerging(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = LayerNorm(4 * dim)

    def forward(self, x, H, W):
        B, L, C = x.shape
        x = x.view(B, H, W, C)
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))  # odd sizes are padded (reference :326-328)
        x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
        return self.reduction(self.norm(x.view(B, -1, 4 * C)))


class BasicLayer(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size, mlp_ratio, drop_path, downsample):
        super().__init__()
        self.window_size, self.shift_size = window_size, window_size // 2
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, num_heads, window_size, 0 if i % 2 == 0 else window_size // 2,
                                 mlp_ratio, drop_path[i]) for i in range(depth)])
        self.downsample = PatchMerging(dim) if downsample else None

    def forward(self, x, H, W, dp=None):
        ws, ss = self.window_size, self.shift_size
        made = []

        def attn_mask():
            """The shifted windows' [nW, ws*ws, ws*ws] additive mask (reference swin_transformer.py:382-398), built when a
            block asks for it: the native window attention does the same bookkeeping by index arithmetic inside its kernel,
            and at the bench size the stage-1 mask is an 11 MB tensor written three times per pass."""
            if not made:
                Hp, Wp = int(math.ceil(H / ws)) * ws, int(math.ceil(W / ws)) * ws
                img_mask = torch.zeros((1, Hp, Wp, 1), device=x.device)
                cnt = 0
                for h in (slice(0, -ws), slice(-ws, -ss), slice(-ss, None)):
                    for w in (slice(0, -ws), slice(-ws, -ss), slice(-ss, None)):
                        img_mask[:, h, w, :] = cnt
                        cnt += 1
                mw = window_partition(img_mask, ws).view(-1, ws * ws)
                m = mw.unsqueeze(1) - mw.unsqueeze(2)
                made.append(m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0))
            return made[0]

        for i, blk in enumerate(self.blocks):
            x = blk(x, H, W, attn_mask, None if dp is None else dp[i])
        if self.downsample is not None:
            return x, H, W, self.downsample(x, H, W), (H + 1) // 2, (W + 1) // 2
        return x, H, W, x, H, W


class PatchEmbed(nn.Module):
    def __init__(self, patch_size=4, in_chans=3, embed_dim=96):
        super().__init__()
        self.patch_size = patch_size
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = LayerNorm(embed_dim)

    def forward(self, x):
        _, _, H, W = x.shape
        p = self.patch_size
        x = F.pad(x, (0, (p - W % p) % p, 0, (p - H % p) % p))  # 1333 -> 1336 (reference :482-489)
        B, C, H, W = x.shape
        Wh, Ww = H // p, W // p
        # 4x4/4 patches as ONE token-major GEMM [B*L, C*p*p] x [C*p*p, E] (+ bias): no NCHW round trip
        # (the conv-shaped result cost a 51 MB transpose in front of the LayerNorm)
        cols = x.view(B, C, Wh, p, Ww, p).permute(0, 2, 4, 1, 3, 5).reshape(B * Wh * Ww, C * p * p)
        w = self.proj.weight.view(self.proj.out_channels, -1)
        x = torch.addmm(self.proj.bias, cols, w.t()) if self.proj.bias is not None else cols @ w.t()
        return self.norm(x.view(B, Wh * Ww, -1)), Wh, Ww


class SwinTransformer(nn.Module):
    def __init__(self, embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window_size=7,
                 mlp_ratio=4.0, drop_path_rate=0.2, out_indices=(1, 2, 3), **_unused):
        super().__init__()
        self.out_indices = tuple(out_indices)
        self.num_layers = len(depths)
        self.patch_embed = PatchEmbed(4, 3, embed_dim)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]
        self._drop_probs = list(dpr)
        self.register_buffer("_keep_probs", 1.0 - torch.tensor(dpr, dtype=torch.float32), persistent=False)
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]
        self.layers = nn.ModuleList([
            BasicLayer(self.num_features[i], depths[i], num_heads[i], window_size, mlp_ratio,
                       dpr[sum(depths[:i]):sum(depths[:i + 1])], downsample=i < self.num_layers - 1)
            for i in range(self.num_layers)])
        for i in self.out_indices:
            self.add_module(f"norm{i}", LayerNorm(self.num_features[i]))

    def _stochastic_depth_scales(self, B, device, dtype):
        """keep / keep_prob factors [blocks, 2, B, 1, 1] of every residual branch, drawn at once (training mode with
        stochastic depth only): same distribution as one bernoulli_ per use (timm DropPath), 4 launches instead of 66."""
        if not self.training or max(self._drop_probs) == 0.0:
            return None
        keep = self._keep_probs.view(-1, 1, 1)              # (a buffer: no host-to-device copy inside graph capture)
        u = torch.rand((keep.shape[0], 2, B), device=device)
        return ((u < keep).to(dtype) / keep.to(dtype)).view(keep.shape[0], 2, B, 1, 1)

    def forward(self, tensor_list: NestedTensor):
        x, Wh, Ww = self.patch_embed(tensor_list.tensors)
        outs = {}
        scales = self._stochastic_depth_scales(x.shape[0], x.device, x.dtype)
        first = 0
        for i, layer in enumerate(self.layers):
            n = len(layer.blocks)
            x_out, H, W, x, Wh, Ww = layer(x, Wh, Ww, None if scales is None else scales[first:first + n])
            first += n
            if i in self.out_indices:
                x_out = getattr(self, f"norm{i}")(x_out)
                out = x_out.view(-1, H, W, self.num_features[i]).permute(0, 3, 1, 2).contiguous()
                m = tensor_list.mask
                mask = F.interpolate(m[None].float(), size=out.shape[-2:]).to(torch.bool)[0]
                outs[len(outs)] = NestedTensor(out, mask)
        return outs


class PositionEmbeddingSineHW(nn.Module):
    """Sine position encoding with separate H/W temperatures (reference position_encoding.py:78-134)."""

    def __init__(self, num_pos_feats=64, temperatureH=10000, temperatureW=10000, normalize=False, scale=None):
        super().__init__()
        if scale is not None and normalize is False:
            raise ValueError("normalize should be True if scale is passed")
        self.num_pos_feats = num_pos_feats
        self.temperatureH, self.temperatureW = temperatureH, temperatureW
        self.normalize = normalize
        self.scale = 2 * math.pi if scale is None else scale

    native = True   # GPU masks: one launch per level (csrc/refpoints.hip zira_sine_pos_hw_f32), bit-identical to the op chain below

    def _dim_t(self, device):
        """temperature ** (2 (i // 2) / F) for both axes, formed with the reference's ops, once per device."""
        cache = self.__dict__.setdefault("_dim_t_cache", {})
        key = str(device)
        if key not in cache:
            i = torch.arange(self.num_pos_feats, dtype=torch.float32, device=device)
            expo = 2 * torch.div(i, 2, rounding_mode="floor") / self.num_pos_feats
            cache[key] = ((self.temperatureH ** expo).contiguous(), (self.temperatureW ** expo).contiguous())
        return cache[key]

    def forward(self, tensor_list: NestedTensor):
        mask = tensor_list.mask
        if self.native and mask.is_cuda and mask.dtype == torch.bool and mask.dim() == 3 and self.num_pos_feats % 2 == 0:
            from . import _lib
            m = mask.contiguous()
            B, H, W = m.shape
            dty, dtx = self._dim_t(m.device)
            out = torch.empty((B, H, W, 2 * self.num_pos_feats), device=m.device, dtype=torch.float32)
            with torch.cuda.device(m.device):
                rc = _lib.load().zira_sine_pos_hw_f32(m.data_ptr(), B, H, W, self.num_pos_feats, int(bool(self.normalize)),
                                                      float(self.scale), 1e-6, dty.data_ptr(), dtx.data_ptr(), out.data_ptr(),
                                                      torch.cuda.current_stream(m.device).cuda_stream)
            if rc != 0:
                raise RuntimeError("zira_sine_pos_hw_f32 failed: hipError %d" % rc)
            return out.permute(0, 3, 1, 2)
        not_mask = ~mask
        y_embed = not_mask.cumsum(1, dtype=torch.float32)
        x_embed = not_mask.cumsum(2, dtype=torch.float32)
        if self.normalize:
            eps = 1e-6
            y_embed = y_embed / (y_embed[:, -1:, :] + eps) * self.scale
            x_embed = x_embed / (x_embed[:, :, -1:] + eps) * self.scale
        i = torch.arange(self.num_pos_feats, dtype=torch.float32, device=mask.device)
        expo = 2 * torch.div(i, 2, rounding_mode="floor") / self.num_pos_feats
        pos_x = x_embed[:, :, :, None] / (self.temperatureW ** expo)
        pos_y = y_embed[:, :, :, None] / (self.temperatureH ** expo)
        pos_x = torch.stack((pos_x[..., 0::2].sin(), pos_x[..., 1::2].cos()), dim=4).flatten(3)
        pos_y = torch.stack((pos_y[..., 0::2].sin(), pos_y[..., 1::2].cos()), dim=4).flatten(3)
        return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


class Joiner(nn.Sequential):
    """(backbone, position_embedding) -> (list of NestedTensor features, list of pos) -- reference
    backbone.py:146-160; ``self[1]`` is also used on its own for the extra 4th level."""

    def __init__(self, backbone, position_embedding):
        super().__init__(backbone, position_embedding)

    def refresh_derived(self):
        """The bf16 planes of the split-bf16 arithmetic (``_frozen_linear``) follow their weights in place: graphs.GraphedNoGrad
        calls this before every replay of the front end's graph, which re-runs no Python."""
        from . import gemm_bf16x3 as g3
        for m in self.modules():
            if m.__dict__.get("_bf16x3_split"):
                g3.refresh(m, {k[0]: getattr(m, k[0]).weight for k in m.__dict__["_bf16x3_split"]})

    def forward(self, tensor_list: NestedTensor):
        xs = self[0](tensor_list)
        out: List[NestedTensor] = []
        pos = []
        for _, x in xs.items():
            out.append(x)
            pos.append(self[1](x).to(x.tensors.dtype))
        return out, pos


def build_backbone(args):
    """Swin variants of reference backbone.py:163-221 (ResNet is not on the ZiRa path)."""
    if args.backbone not in SWIN_VARIANTS:
        raise NotImplementedError("Unknown backbone {}".format(args.backbone))
    pos = PositionEmbeddingSineHW(args.hidden_dim // 2, temperatureH=args.pe_temperatureH,
                                  temperatureW=args.pe_temperatureW, normalize=True)
    idx = list(args.return_interm_indices)
    assert idx in [[0, 1, 2, 3], [1, 2, 3], [3]]
    swin = SwinTransformer(out_indices=tuple(idx), **SWIN_VARIANTS[args.backbone])
    model = Joiner(swin, pos)
    model.num_channels = swin.num_features[4 - len(idx):]
    return model
