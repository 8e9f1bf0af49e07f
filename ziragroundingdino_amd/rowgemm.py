"""Host side of ``zira_rowgemm_f32`` (csrc/rowgemm.hip): the nn.Linear calls of a decoder layer on its B x 900 query rows with
the position-code add in front of them and the residual add + LayerNorm behind them in the same launch, and the same for
their input gradients (reference transformer_for_adapter.py:1001-1071; ms_deform_attn.py:262-288, :338).

No autograd here: the callers are the hand-written forward / backward of ``decoder_layer._FrozenDecoderLayer``.
There is no fallback: without the HIP library the call raises."""
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib


def supported(m: int, n: int, k: int, layer_norm: bool = False) -> bool:
    """Shapes the kernel takes (include/zira_msda.h): K in multiples of 128 up to 2048, N in multiples of 128 (== 256 with
    the LayerNorm epilogue)."""
    return m >= 0 and k % 128 == 0 and 0 < k <= 2048 and (n == 256 if layer_norm else (n > 0 and n % 128 == 0))


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _rows(t: Tensor) -> Tensor:
    """[..., C] float32 tensor as contiguous-row 2-D (a view; row-strided slices of wider tensors keep their stride)."""
    assert t.dtype == torch.float32 and t.stride(-1) == 1
    if t.dim() == 2:
        return t
    assert t.is_contiguous(), "multi-dimensional operands must be contiguous"
    return t.view(-1, t.shape[-1])


def rowgemm(a: Tensor, w: Tensor, *, w_is_nk: bool, n: Optional[int] = None, bias: Optional[Tensor] = None,
            pos: Optional[Tensor] = None, pos_cols: int = 0, res: Optional[Tensor] = None, mask: Optional[Tensor] = None,
            relu: bool = False, ln: Optional[Tuple[Tensor, Tensor, float]] = None, ln_save: bool = False,
            lnb: Optional[Tuple[Tensor, Tensor, Tensor, Tensor]] = None, lnb_save: bool = False,
            batch: int = 0, a_batch_first: bool = False, c_batch_first: bool = False, out: Optional[Tensor] = None):
    """``epilogue(prologue(a) @ op(w))`` on the rows of ``a``; see include/zira_msda.h for the pieces.

    a [M, K] (or [..., K] contiguous); w [N, K] with ``w_is_nk`` (x W^T) else [K, N] (g W).  ``ln = (gamma, beta, eps)``
    normalises the rows (``ln_save``: also return the rows before the LayerNorm and (mean, rstd)); ``lnb = (x, gamma, mean,
    rstd)`` makes the operand the LayerNorm input gradient of ``a`` (``lnb_save``: also return it).  Returns ``c`` or
    ``(c, extras...)`` in the order ln_sum, mean, rstd, lnb_dx."""
    a2 = _rows(a)
    m, k = a2.shape
    w2 = _rows(w)
    if n is None:
        n = w2.shape[0] if w_is_nk else w2.shape[1]
    assert (w2.shape == (n, k)) if w_is_nk else (w2.shape[0] == k and w2.shape[1] >= n), (tuple(w2.shape), n, k, w_is_nk)
    dev = a2.device
    if m == 0 and not (ln_save or lnb_save):
        return out if out is not None else torch.empty((0, n), device=dev, dtype=torch.float32)
    args = _lib.RowGemmArgs()
    args.a, args.lda = a2.data_ptr(), a2.stride(0)
    if pos is not None:
        p2 = _rows(pos)
        assert p2.shape == (m, k)
        args.pos, args.ldpos, args.pos_cols = p2.data_ptr(), p2.stride(0), (pos_cols or n)
    args.w, args.ldw, args.w_is_nk = w2.data_ptr(), w2.stride(0), int(w_is_nk)
    args.bias = _ptr(bias)
    if res is not None:
        r2 = _rows(res)
        assert r2.shape == (m, n)
        args.res, args.ldres = r2.data_ptr(), r2.stride(0)
    if mask is not None:
        assert mask.is_contiguous() and mask.numel() == m * n
        args.mask = mask.data_ptr()
    args.relu = int(relu)
    c = out if out is not None else torch.empty((m, n), device=dev, dtype=torch.float32)
    c2 = _rows(c)
    assert c2.shape == (m, n)
    extras = []
    if ln is not None:
        gamma, beta, eps = ln
        args.ln_gamma, args.ln_beta, args.ln_eps = gamma.data_ptr(), _ptr(beta), float(eps)
        if ln_save:
            ln_sum = torch.empty((m, n), device=dev, dtype=torch.float32)
            mean = torch.empty(m, device=dev, dtype=torch.float32)
            rstd = torch.empty(m, device=dev, dtype=torch.float32)
            args.ln_sum, args.ln_mean, args.ln_rstd = ln_sum.data_ptr(), mean.data_ptr(), rstd.data_ptr()
            extras += [ln_sum, mean, rstd]
    if lnb is not None:
        x, gamma, mean_b, rstd_b = lnb
        assert x.is_contiguous() and x.numel() == m * k
        args.lnb_x, args.lnb_gamma = x.data_ptr(), _ptr(gamma)
        args.lnb_mean, args.lnb_rstd = mean_b.data_ptr(), rstd_b.data_ptr()
        if lnb_save:
            dx = torch.empty((m, k), device=dev, dtype=torch.float32)
            args.lnb_dx = dx.data_ptr()
            extras.append(dx)
    args.c, args.ldc = c2.data_ptr(), c2.stride(0)
    args.m, args.n, args.k = m, n, k
    args.batch, args.a_batch_first, args.c_batch_first = int(batch), int(a_batch_first), int(c_batch_first)
    rc = _lib.load().zira_rowgemm_f32(args, torch.cuda.current_stream(dev).cuda_stream)
    if rc != 0:
        raise RuntimeError("zira_rowgemm_f32 failed with code %d (M=%d N=%d K=%d)" % (rc, m, n, k))
    return (c, *extras) if extras else c
