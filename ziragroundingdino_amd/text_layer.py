"""The Linear layers of a frozen text-enhancer layer (reference transformer_vanilla.py:72-123: post-LN self-attention over
the <= 256 text tokens under the sub-sentence mask, then an FFN) through the row GEMMs of csrc/rowgemm.hip.

On 2 x 32 tokens every GEMM of the layer is a few MFLOP that the library runs in 10-20 us, and between them sit the position
add, two residual adds, two LayerNorms, a ReLU and their backward passes: 43 launches, 267 us per layer forward + backward.
With the row-GEMM family the layer is three autograd nodes round the attention core, which stays with PyTorch (head width 64
and a full [T, T] mask are outside csrc/attn.hip):

    qkv  = (src + pos | src + pos | src) W_in^T + b_in              position add as the GEMM's prologue
    src1 = LayerNorm(src + attn W_out^T + b_out)                    bias, residual and LayerNorm as its epilogue
    out  = LayerNorm(src1 + relu(src1 W1^T + b1) W2^T + b2)         ReLU / bias, residual and LayerNorm epilogues

and in the backward the LayerNorm gradient is the prologue of the GEMM that follows it, the ReLU mask and the residual
gradient its epilogue.  Frozen fp32 weights on the GPU without dropout only; otherwise the modules run."""
import torch
from torch.autograd.function import once_differentiable

from . import rowgemm as rg
from .rowgemm import rowgemm


def applies(layer, src, pos) -> bool:
    sa = layer.self_attn
    if not (src.is_cuda and src.dtype == torch.float32 and src.dim() == 3 and not torch.is_autocast_enabled("cuda")):
        return False
    if layer.normalize_before or layer.activation is not torch.nn.functional.relu:
        return False
    if layer.training and (layer.dropout.p > 0 or layer.dropout1.p > 0 or layer.dropout2.p > 0 or sa.dropout > 0):
        return False
    if not (sa._qkv_same_embed_dim and sa.bias_k is None and not sa.add_zero_attn and not sa.batch_first
            and sa.in_proj_bias is not None and sa.out_proj.bias is not None):
        return False
    ps = [sa.in_proj_weight, sa.in_proj_bias, sa.out_proj.weight, sa.out_proj.bias, layer.linear1.weight, layer.linear1.bias,
          layer.linear2.weight, layer.linear2.bias, layer.norm1.weight, layer.norm1.bias, layer.norm2.weight, layer.norm2.bias]
    if any(p is None or p.requires_grad or p.dtype != torch.float32 or p.device != src.device for p in ps):
        return False
    if pos is not None and (pos.requires_grad or pos.shape != src.shape or pos.dtype != torch.float32):
        return False
    E, F_ = src.shape[-1], layer.linear1.out_features
    rows = src.shape[0] * src.shape[1]
    return (E == 256 and rg.supported(rows, 3 * E, E) and rg.supported(rows, E, E, layer_norm=True)
            and rg.supported(rows, F_, E) and rg.supported(rows, E, F_, layer_norm=True))


def _weights(layer):
    """[K, N] copies of the four weights (the row GEMM streams W rows: [N, K] costs four times the cache lines), rebuilt in
    place -- captured graphs keep reading them -- when a weight's version changes (load_state_dict, .to())."""
    sa = layer.self_attn
    ws = (sa.in_proj_weight, sa.out_proj.weight, layer.linear1.weight, layer.linear2.weight)
    key = tuple(x for w in ws for x in (w.data_ptr(), w._version))
    cached = getattr(layer, "_text_layer_wt", None)
    if cached is None or cached[0] != key:
        with torch.no_grad():
            if cached is not None and all(o.shape == (w.shape[1], w.shape[0]) and o.device == w.device for o, w in zip(cached[1], ws)):
                for o, w in zip(cached[1], ws):
                    o.copy_(w.t())
                new = cached[1]
            else:
                new = [w.detach().t().contiguous() for w in ws]
        cached = layer._text_layer_wt = (key, new)
    return cached[1]


class _QKV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pos, w_t, bias, w):
        ctx.save_for_backward(w)
        E = x.shape[1]
        return rowgemm(x, w_t, w_is_nk=False, bias=bias, pos=pos, pos_cols=2 * E)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        return rowgemm(g.contiguous(), w, w_is_nk=False), None, None, None, None


class _ProjAddNorm(torch.autograd.Function):
    """LayerNorm(res + a W^T + b); backward: (da, dres) with the LayerNorm gradient as the GEMM's prologue."""

    @staticmethod
    def forward(ctx, a, res, w_t, bias, w, gamma, beta, eps):
        out, s, mean, rstd = rowgemm(a, w_t, w_is_nk=False, bias=bias, res=res, ln=(gamma, beta, eps), ln_save=True)
        ctx.save_for_backward(w, s, mean, rstd, gamma)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        w, s, mean, rstd, gamma = ctx.saved_tensors
        da, ds = rowgemm(g.contiguous(), w, w_is_nk=False, lnb=(s, gamma, mean, rstd), lnb_save=True)
        return (da, ds) + (None,) * 6


class _FFNAddNorm(torch.autograd.Function):
    """LayerNorm(x + relu(x W1^T + b1) W2^T + b2)."""

    @staticmethod
    def forward(ctx, x, w1_t, b1, w2_t, b2, w1, w2, gamma, beta, eps):
        h = rowgemm(x, w1_t, w_is_nk=False, bias=b1, relu=True)
        out, s, mean, rstd = rowgemm(h, w2_t, w_is_nk=False, bias=b2, res=x, ln=(gamma, beta, eps), ln_save=True)
        ctx.save_for_backward(w1, w2, h, s, mean, rstd, gamma)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        w1, w2, h, s, mean, rstd, gamma = ctx.saved_tensors
        gh, ds = rowgemm(g.contiguous(), w2, w_is_nk=False, mask=h, lnb=(s, gamma, mean, rstd), lnb_save=True)
        return (rowgemm(gh, w1, w_is_nk=False, res=ds),) + (None,) * 9


def forward(layer, src, pos, src_mask):
    """The layer's forward (reference transformer_vanilla.py:100-123); call only when ``applies()``."""
    from .transformer import _mha_core
    sa = layer.self_attn
    T, B, E = src.shape
    w_in_t, w_out_t, w1_t, w2_t = _weights(layer)
    x = src.contiguous().view(T * B, E)
    p = None if pos is None else pos.contiguous().view(T * B, E)
    if p is None:
        p = torch.zeros_like(x)
    qkv = _QKV.apply(x, p, w_in_t, sa.in_proj_bias, sa.in_proj_weight).view(T, B, 3 * E)
    q, k, v = qkv.split(E, dim=-1)    # (split: its backward is one cat)
    a = _mha_core(q, k, v, sa.num_heads, None, src_mask, 0.0)                               # [T, B, E]
    n1, n2 = layer.norm1, layer.norm2
    src1 = _ProjAddNorm.apply(a.view(T * B, E), x, w_out_t, sa.out_proj.bias, sa.out_proj.weight, n1.weight, n1.bias, n1.eps)
    out = _FFNAddNorm.apply(src1, w1_t, layer.linear1.bias, w2_t, layer.linear2.bias, layer.linear1.weight, layer.linear2.weight,
                            n2.weight, n2.bias, n2.eps)
    return out.view(T, B, E)
