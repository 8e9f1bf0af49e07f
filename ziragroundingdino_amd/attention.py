"""Fused small attention on the GPU (C ABI ``zira_attn_{fwd,bwd}_f32``, csrc/attn.hip): what
``nn.MultiheadAttention`` computes between its projections for the decoder's self-attention over the queries and its
cross-attention to the text tokens (reference transformer_for_adapter.py:1043-1058), forward and backward, with the
scores kept in registers.  fp32, head width 32, no dropout, optional additive key-padding mask."""
import math

import torch

from . import _lib


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _row_stride(t):
    """Floats between consecutive (l, b) rows of a [rows, B, E] view whose last dimension is contiguous, or None."""
    L, B, E = t.shape
    if t.stride(2) != 1 or t.stride(0) != B * t.stride(1) or t.stride(1) < E:
        return None
    if t.stride(1) % 4 or t.data_ptr() % 16:
        return None
    return t.stride(1)


def supported(q, k, v, num_heads, key_mask=None):
    if not (q.is_cuda and q.dtype == torch.float32 and k.dtype == torch.float32 and v.dtype == torch.float32):
        return False
    if q.dim() != 3 or q.shape[2] != num_heads * 32 or k.shape != v.shape or k.shape[1:] != q.shape[1:]:
        return False
    if q.shape[1] * num_heads > 65535:
        return False
    if key_mask is not None and (key_mask.dtype != torch.float32 or tuple(key_mask.shape) != (q.shape[1], k.shape[0])):
        return False
    return True


class _FusedAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, key_mask, num_heads):
        lib = _lib.load()
        q, k, v = (t if _row_stride(t) is not None else t.contiguous() for t in (q, k, v))
        L, B, E = q.shape
        S = k.shape[0]
        km = key_mask.contiguous() if key_mask is not None else None
        out = torch.empty((L, B, E), dtype=torch.float32, device=q.device)
        lse = torch.empty((B, num_heads, L), dtype=torch.float32, device=q.device)
        scale = 1.0 / math.sqrt(32.0)
        with torch.cuda.device(q.device):
            rc = lib.zira_attn_fwd_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), km.data_ptr() if km is not None else None,
                                       L, S, B, num_heads, 32, _row_stride(q), _row_stride(k), _row_stride(v), scale,
                                       out.data_ptr(), lse.data_ptr(), _stream())
        if rc != 0:
            raise RuntimeError("zira_attn_fwd_f32 failed with code %d" % rc)
        ctx.save_for_backward(q, k, v, km, out, lse)
        ctx.num_heads = num_heads
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, km, out, lse = ctx.saved_tensors
        lib = _lib.load()
        H = ctx.num_heads
        L, B, E = q.shape
        S = k.shape[0]
        dout = dout.contiguous()
        dq = torch.empty((L, B, E), dtype=torch.float32, device=q.device)
        dk = torch.empty((S, B, E), dtype=torch.float32, device=q.device)
        dv = torch.empty((S, B, E), dtype=torch.float32, device=q.device)
        nscr = lib.zira_attn_bwd_scratch_floats(L, S, B, H)
        scratch = torch.empty((nscr,), dtype=torch.float32, device=q.device)
        with torch.cuda.device(q.device):
            rc = lib.zira_attn_bwd_f32(q.data_ptr(), k.data_ptr(), v.data_ptr(), km.data_ptr() if km is not None else None,
                                       out.data_ptr(), dout.data_ptr(), lse.data_ptr(), L, S, B, H, 32,
                                       _row_stride(q), _row_stride(k), _row_stride(v), 1.0 / math.sqrt(32.0),
                                       dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), scratch.data_ptr(), nscr, _stream())
        if rc != 0:
            raise RuntimeError("zira_attn_bwd_f32 failed with code %d" % rc)
        return dq, dk, dv, None, None


def fused_attention(q, k, v, num_heads, key_mask=None):
    """q [L, B, H*32], k / v [S, B, H*32] (views with a contiguous last dimension are taken in place), ``key_mask``
    additive [B, S] or None -> softmax(q k^T / sqrt(32) + mask) v as [L, B, H*32]."""
    return _FusedAttention.apply(q, k, v, key_mask, num_heads)
