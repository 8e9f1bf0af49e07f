"""Convolutions of the hot path expressed as GEMMs.

On this ROCm build MIOpen has no tuned fp32 solver for the 1x1 / patch / 3x3-stride-2
convolutions of the input projections and falls back to ``naive_conv_*`` kernels (3-4 ms per
call, ~25 % of the kernel time of a training step in the first rocprofv3 trace).  Every one of
them is a plain dense contraction, so they are routed to the GEMM library (hipBLASLt, MFMA)
instead: 1x1 -> W[O,C] @ x[B,C,HW]; kernel == stride (patch embedding) -> reshape + GEMM;
anything else -> im2col (F.unfold) + GEMM.  Same results as F.conv2d up to fp32 summation
order; gradients come from autograd of the matmuls (weight gradients are GEMMs as well).
"""
import torch
from torch.autograd.function import once_differentiable
import torch.nn.functional as F


def _scratch(cache, dev, n):
    """n floats of kernel scratch: cached per (device, stream) in eager mode; while a stream is being
    captured into a hipGraph the allocation is left to the graph's own pool (a cached tensor would be
    baked into the graph and then replaced or shared behind its back)."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(n, dtype=torch.float32, device=dev)
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = cache.get(key)
    if ws is None or ws.numel() < n:
        ws = cache[key] = torch.empty(n, dtype=torch.float32, device=dev)
    return ws


def conv_columns(x, kernel, stride=1, padding=0):
    """im2col view of x for a (kh, kw) kernel: ([B, C*kh*kw, Ho*Wo], Ho, Wo); free for 1x1."""
    kh, kw = kernel
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (padding, padding) if isinstance(padding, int) else padding
    B, C, H, W = x.shape
    if kh == 1 and kw == 1 and sh == 1 and sw == 1 and ph == 0 and pw == 0:
        return x.flatten(2), H, W
    if kh == sh and kw == sw and ph == 0 and pw == 0 and H % kh == 0 and W % kw == 0:
        Ho, Wo = H // kh, W // kw                                            # non-overlapping patches
        cols = x.view(B, C, Ho, kh, Wo, kw).permute(0, 1, 3, 5, 2, 4).reshape(B, C * kh * kw, Ho * Wo)
        return cols, Ho, Wo
    Ho = (H + 2 * ph - kh) // sh + 1
    Wo = (W + 2 * pw - kw) // sw + 1
    return F.unfold(x, (kh, kw), padding=(ph, pw), stride=(sh, sw)), Ho, Wo


def conv2d_as_gemm(x, weight, bias=None, stride=1, padding=0):
    """conv2d (dilation 1, groups 1, zero padding) via GEMM.  x [B,C,H,W], weight [O,C,kh,kw]."""
    cols, Ho, Wo = conv_columns(x, weight.shape[2:], stride, padding)
    return _columns_gemm(cols, weight, bias).view(x.shape[0], weight.shape[0], Ho, Wo)


def _columns_gemm(cols, weight, bias):
    """weight [O, ...] applied to im2col columns [B, K, N] -> [B, O, N].  bmm with the weight expanded over the batch:
    matmul(2-D, 3-D) transposes and clones the columns first (25 MB at the first level); the bias joins in place."""
    O = weight.shape[0]
    y = torch.bmm(weight.view(1, O, -1).expand(cols.shape[0], -1, -1), cols)
    if bias is not None:
        y = y.add_(bias.view(1, O, 1))
    return y


def conv2d_pair_as_gemm(x, weight_a, bias_a, weight_b, bias_b, stride=1, padding=0):
    """Two convolutions of the same input with equally shaped kernels; returns two contiguous [B,O,Ho,Wo].
    (Round 1 ran them as ONE batched GEMM over the stacked weights; the broadcast made matmul clone the columns for both
    weight sets, the results had to be selected out of the stacked tensor -- two zero-filled 2x-sized gradients in the
    backward -- and the bias was a pass over both: two plain GEMMs on the same columns move less.)"""
    cols, Ho, Wo = conv_columns(x, weight_a.shape[2:], stride, padding)
    shape = (x.shape[0], weight_a.shape[0], Ho, Wo)
    return _columns_gemm(cols, weight_a, bias_a).view(shape), _columns_gemm(cols, weight_b, bias_b).view(shape)


def conv_module_as_gemm(conv: torch.nn.Conv2d, x):
    """Apply an nn.Conv2d's parameters through conv2d_as_gemm (falls back to the module itself
    for dilation / groups / non-zero padding modes, which the hot path does not use)."""
    if conv.dilation != (1, 1) or conv.groups != 1 or conv.padding_mode != "zeros" or isinstance(conv.padding, str):
        return conv(x)
    return conv2d_as_gemm(x, conv.weight, conv.bias, conv.stride, conv.padding)


# ---------------------------------------------------------------------------------------------
# Tall-reduction products (csrc/xty.hip): out[z] = X[z]^T @ Y[z] with a reduction over tens of
# thousands of image tokens and a small output -- the shape BiMultiHeadAttention's re-bracketed
# image side produces (transformer.py).  rocBLAS gives them one or two tiles (109 / 85 us at
# N = 22223, 64 x 256); the split-reduction kernel takes ~15 us (128 x 256 / 256 x 128, the bench's shapes: 36 us, xty_rows128).
# ---------------------------------------------------------------------------------------------
_XTY_WS = {}


def _use_xty(X, Y, a, b, N):
    return (X.is_cuda and X.dtype == torch.float32 and Y.dtype == torch.float32 and N >= 2048
            and a % 4 == 0 and b % 4 == 0 and a * b <= 512 * 512)


def _xty_native(X, Y, x_transposed):
    from . import _lib

    lib = _lib.load()
    X, Y = X.contiguous(), Y.contiguous()
    B, N, b = Y.shape
    a = X.shape[1] if x_transposed else X.shape[2]
    out = torch.empty((B, a, b), dtype=torch.float32, device=X.device)
    n = lib.zira_xty_workspace_floats(B, N, a, b)
    ws = _scratch(_XTY_WS, X.device, n)
    rc = lib.zira_xty_f32(X.data_ptr(), Y.data_ptr(), B, N, a, b, int(bool(x_transposed)), out.data_ptr(),
                          ws.data_ptr(), torch.cuda.current_stream(X.device).cuda_stream)
    if rc != 0:
        raise RuntimeError("zira_xty_f32 failed with HIP error %d" % rc)
    return out


TALL_BF16X3_MIN_COLS = 192
USE_TALL_BF16X3 = True    # developer switch (scripts/ab_step.py tall=0): csrc/xty.hip (fp32 matrix instruction) instead
_TALL_WS = {}


def _use_tall_bf16x3(X, Y, a, b, N, x_transposed):
    """Both operands token-major, one of them 256 wide and the other at least TALL_BF16X3_MIN_COLS, under the split arithmetics
    (csrc/xty_bf16x3.hip against the fp32 matrix instruction of csrc/xty.hip at 22223 tokens x 2 images: 27.8 / 28.8 us at 64
    columns, 36.7 / 41.0 at 128, 88 / 210 at 388, 154 / 399 at 776; inside the training step the 128-column calls did not repay
    it -- 24.10 against 24.03 ms per step -- so the narrow ones stay with csrc/xty.hip)."""
    from . import gemm_bf16x3 as g3
    n = b if a == 256 else a
    return (USE_TALL_BF16X3 and not x_transposed and g3.enabled() and (a == 256 or b == 256) and a % 4 == 0 and b % 4 == 0
            and TALL_BF16X3_MIN_COLS <= n <= 2048 and X.shape[0] <= 64 and not torch.is_autocast_enabled("cuda"))


def _tall_bf16x3(X, Y):
    from . import _lib

    lib = _lib.load()
    X, Y = X.contiguous(), Y.contiguous()
    B, N, a = X.shape
    b = Y.shape[2]
    out = torch.empty((B, a, b), dtype=torch.float32, device=X.device)
    thin_first = b == 256                       # P = the operand that is not (necessarily) 256 wide
    P, Q, n = (X, Y, a) if thin_first else (Y, X, b)
    ws = _scratch(_TALL_WS, X.device, lib.zira_xty_bf16x3_workspace_floats(B, N, n))
    with torch.cuda.device(X.device):
        rc = lib.zira_xty_bf16x3_f32(P.data_ptr(), Q.data_ptr(), B, N, n, 0 if thin_first else 1, out.data_ptr(), ws.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError("zira_xty_bf16x3_f32 failed with code %d" % rc)
    return out


def xty(X, Y, x_transposed=False):
    """X^T @ Y per batch element, no autograd: X [B, N, a] (or [B, a, N] with ``x_transposed``),
    Y [B, N, b] -> [B, a, b]."""
    N, b = Y.shape[1], Y.shape[2]
    a = X.shape[1] if x_transposed else X.shape[2]
    if _use_xty(X, Y, a, b, N):
        if _use_tall_bf16x3(X, Y, a, b, N, x_transposed):
            return _tall_bf16x3(X, Y)
        return _xty_native(X, Y, x_transposed)
    return torch.bmm(X if x_transposed else X.transpose(1, 2), Y)


# ---- thin products of the fusion block's image side (csrc/thin_f16x2.hip) ----------------------
USE_THIN = True          # developer switch (scripts/ab_step.py thin=0): the library's bmm / the row GEMM instead
THIN_MIN_ROWS = 2048


def thin_supported(A, N, K, res=None):
    """A [B, M, K] fp32 on the GPU against a [K, N] matrix per image: K <= 128 or K == 256 (the panel of 32 rows and all of K
    fits LDS four times per CU), both multiples of 4."""
    from . import gemm_bf16x3 as g3
    return (USE_THIN and g3.enabled() and A.is_cuda and A.dtype == torch.float32 and A.dim() == 3 and A.shape[1] >= THIN_MIN_ROWS
            and A.shape[0] <= 64 and K % 4 == 0 and N % 4 == 0 and (K <= 128 or K == 256) and N <= 2048
            and (res is None or res.dtype == torch.float32) and not torch.is_autocast_enabled("cuda"))


def _thin_frags(lib, W, w_is_kn, B, N, K, stream):
    W = W.contiguous()
    assert W.shape == ((B, K, N) if w_is_kn else (B, N, K)) and W.dtype == torch.float32
    fb = lib.zira_thin_f16x2_frag_bytes(N, K)
    frags = torch.empty(B * fb, dtype=torch.uint8, device=W.device)
    rc = lib.zira_thin_f16x2_split_f32(W.data_ptr(), B, N, K, int(bool(w_is_kn)), frags.data_ptr(), stream)
    if rc != 0:
        raise RuntimeError("zira_thin_f16x2_split_f32 failed with code %d" % rc)
    return frags


def thin_bmm(A, W, w_is_kn, A2=None, W2=None, bias=None, res=None, out=None, w2_is_kn=None):
    """A [B, M, K] @ W (+ A2 @ W2) (+ bias [B, N]) (+ res [B, M, N]) -> [B, M, N], no autograd; W [B, K, N] (``w_is_kn``) or
    [B, N, K] (then it is W^T that multiplies); ``w2_is_kn`` for W2 (default: as W).  Call only when ``thin_supported``."""
    from . import _lib

    lib = _lib.load()
    A = A.contiguous()
    B, M, K = A.shape
    N = W.shape[2] if w_is_kn else W.shape[1]
    with torch.cuda.device(A.device):
        stream = torch.cuda.current_stream().cuda_stream
        f1 = _thin_frags(lib, W, w_is_kn, B, N, K, stream)
        f2 = None
        if A2 is not None:
            A2 = A2.contiguous()
            assert A2.shape == A.shape and A2.dtype == torch.float32
            f2 = _thin_frags(lib, W2, w_is_kn if w2_is_kn is None else w2_is_kn, B, N, K, stream)
        if bias is not None:
            bias = bias.contiguous()
            assert bias.shape == (B, N) and bias.dtype == torch.float32
        if res is not None:
            res = res.contiguous()
            assert res.shape == (B, M, N)
        if out is None:
            out = torch.empty((B, M, N), dtype=torch.float32, device=A.device)
        else:
            assert out.shape == (B, M, N) and out.is_contiguous() and out.dtype == torch.float32
        rc = lib.zira_thin_f16x2_f32(A.data_ptr(), f1.data_ptr(), A2.data_ptr() if A2 is not None else None,
                                     f2.data_ptr() if f2 is not None else None, B, M, N, K,
                                     bias.data_ptr() if bias is not None else None, res.data_ptr() if res is not None else None,
                                     out.data_ptr(), stream)
    if rc != 0:
        raise RuntimeError("zira_thin_f16x2_f32 failed with code %d" % rc)
    thin_bmm.calls += 1
    return out


thin_bmm.calls = 0


def wide_k_supported(A, N, K):
    """The products with the LONG text side on the contraction index (K = H T > 256: COCO-length captions), one tiled two-plane
    GEMM per image (csrc/gemm_f16x2.hip; K a multiple of 4, the small operand split per call)."""
    from . import gemm_bf16x3 as g3
    return (USE_THIN and g3._two_plane() and A.is_cuda and A.dtype == torch.float32 and A.dim() == 3 and A.shape[1] >= THIN_MIN_ROWS
            and A.shape[0] <= 8 and K > 256 and K % 4 == 0 and N % 32 == 0 and not torch.is_autocast_enabled("cuda"))


def wide_k_bmm(A, W, w_is_kn, bias=None, res=None):
    """A [B, M, K] @ W (+ bias [B, N]) (+ res [B, M, N]), W [B, K, N] (``w_is_kn``) or [B, N, K]; call when ``wide_k_supported``."""
    from . import gemm_bf16x3 as g3
    A = A.contiguous()
    B, M, K = A.shape
    N = W.shape[2] if w_is_kn else W.shape[1]
    out = torch.empty((B, M, N), dtype=torch.float32, device=A.device)
    if res is not None:
        res = res.contiguous()
    for i in range(B):
        planes = g3.split_planes_f16x2(W[i].contiguous(), bool(w_is_kn))
        bi = bias[i].contiguous() if bias is not None else g3._zeros(N, A.device)
        if res is None:
            g3.gemm_f16x2(A[i], planes, N, g3.EPI_BIAS, bias=bi, out=out[i])
        else:
            g3.gemm_f16x2(A[i], planes, N, g3.EPI_BIAS_RES, bias=bi, aux=res[i], out=out[i])
    wide_k_bmm.calls += 1
    return out


wide_k_bmm.calls = 0


def _bmm_nn(A, W):
    """A [B, M, K] @ W [B, K, N]"""
    if thin_supported(A, W.shape[2], W.shape[1]):
        return thin_bmm(A, W, True)
    if wide_k_supported(A, W.shape[2], W.shape[1]):
        return wide_k_bmm(A, W, True)
    return torch.bmm(A, W)


def _bmm_nt(A, W):
    """A [B, M, K] @ W [B, N, K]^T"""
    if thin_supported(A, W.shape[1], W.shape[2]):
        return thin_bmm(A, W, False)
    if wide_k_supported(A, W.shape[1], W.shape[2]):
        return wide_k_bmm(A, W, False)
    return torch.bmm(A, W.transpose(1, 2))


class _TallReduce(torch.autograd.Function):
    """P [B, a, N] @ V [B, N, b] -> [B, a, b]  (the forward itself is the tall reduction)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)   # fp32 also under autocast
    def forward(ctx, P, V):
        ctx.save_for_backward(P, V)
        return xty(P, V, x_transposed=True)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        P, V = ctx.saved_tensors
        g = g.to(P.dtype)
        gP = torch.bmm(g, V.transpose(1, 2)) if ctx.needs_input_grad[0] else None
        gV = torch.bmm(P.transpose(1, 2), g) if ctx.needs_input_grad[1] else None
        return gP, gV


class _WideMatmul(torch.autograd.Function):
    """(bias +) L [B, N, k] @ R [B, k, m] -> [B, N, m]: an ordinary GEMM forward whose gradient
    w.r.t. the small right operand, L^T @ g, is the tall reduction."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, L, R, bias):
        ctx.save_for_backward(L, R)
        ctx.has_bias = bias is not None
        if bias is None:
            return _bmm_nn(L, R)
        if bias.dim() == 1 and thin_supported(L, R.shape[2], R.shape[1]):
            return thin_bmm(L, R, True, bias=bias.view(1, -1).expand(L.shape[0], -1))
        if bias.dim() == 1 and wide_k_supported(L, R.shape[2], R.shape[1]):
            return wide_k_bmm(L, R, True, bias=bias.view(1, -1).expand(L.shape[0], -1))
        if L.is_cuda and L.dim() == 3 and L.shape[0] <= 4 and bias.dim() == 1:
            # baddbmm first broadcasts the bias into the [B, N, m] result (a 45 MB copy at the encoder's size) and then reads
            # it back as the GEMM's addend; per image the bias rides in the GEMM epilogue instead
            out = L.new_empty((L.shape[0], L.shape[1], R.shape[2]))
            for i in range(L.shape[0]):
                torch.addmm(bias, L[i], R[i], out=out[i])
            return out
        return torch.baddbmm(bias, L, R)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        L, R = ctx.saved_tensors
        g = g.to(L.dtype).contiguous()
        gL = _bmm_nt(g, R) if ctx.needs_input_grad[0] else None
        gR = xty(L, g) if ctx.needs_input_grad[1] else None
        gb = g.sum((0, 1)) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gL, gR, gb


def tall_reduce(P, V):
    """P [B, a, N] @ V [B, N, b] with autograd."""
    return _TallReduce.apply(P, V)


def wide_matmul(L, R, bias=None):
    """(bias +) L [B, N, k] @ R [B, k, m] with autograd; see ``_WideMatmul``."""
    return _WideMatmul.apply(L, R, bias)


class _WideMatmulResidual(torch.autograd.Function):
    """res + scale * (L @ R + bias): the image-side output of the fusion block with its layer-scale residual
    (reference fuse_modules.py:300-303: ``v + drop_path(gamma_v * delta_v)``) as one row-GEMM launch per image.

    ``scale`` ([m] or [B, 1, m]: the layer scale, times the per-sample stochastic-depth factor) is constant here
    (frozen), so it is folded into the SMALL operands -- R [B, k, m] and the bias -- and the GEMM's epilogue adds the
    residual rows: no [B, N, m] product in memory, no scale pass over it in the forward (2 addmm + addcmul 91 us ->
    64 us at N = 22223, k = 128) and none over its gradient in the backward (the incoming gradient IS the scaled
    product's; only the small dR is scaled back)."""

    @staticmethod
    def forward(ctx, L, R, bias, res, scale):
        from .rowgemm import rowgemm
        B = L.shape[0]
        sc = scale if scale.dim() == 3 else scale.view(1, 1, -1)
        Rs = (R * sc).contiguous()                                    # [B, k, m]
        bs = (bias.view(1, -1) * sc.reshape(-1, sc.shape[-1])).expand(B, -1).contiguous()   # [B, m]
        L, res = L.contiguous(), res.contiguous()
        if thin_supported(L, Rs.shape[2], Rs.shape[1], res):
            out = thin_bmm(L, Rs, True, bias=bs, res=res)
        elif wide_k_supported(L, Rs.shape[2], Rs.shape[1]):
            out = wide_k_bmm(L, Rs, True, bias=bs, res=res)
        else:
            out = torch.empty_like(res)
            for i in range(B):
                rowgemm(L[i], Rs[i], w_is_nk=False, bias=bs[i], res=res[i], out=out[i])
        ctx.save_for_backward(L, Rs, sc)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        L, Rs, sc = ctx.saved_tensors
        g = g.contiguous()
        gL = _bmm_nt(g, Rs) if ctx.needs_input_grad[0] else None
        gR = xty(L, g) * sc if ctx.needs_input_grad[1] else None
        gb = (g * sc).sum((0, 1)) if ctx.needs_input_grad[2] else None
        return gL, gR, gb, (g if ctx.needs_input_grad[3] else None), None


def wide_matmul_residual_supported(L, R, bias, res, scale) -> bool:
    from . import rowgemm as rg
    if not (L.is_cuda and L.dim() == 3 and R.dim() == 3 and res.dim() == 3 and bias is not None and bias.dim() == 1):
        return False
    if any(t.dtype != torch.float32 for t in (L, R, bias, res, scale)) or torch.is_autocast_enabled("cuda"):
        return False
    if scale.requires_grad or L.shape[0] > 4 or res.shape != (L.shape[0], L.shape[1], R.shape[2]):
        return False
    return (thin_supported(L, R.shape[2], R.shape[1], res) or wide_k_supported(L, R.shape[2], R.shape[1])
            or rg.supported(L.shape[1], R.shape[2], L.shape[2]))


def wide_matmul_residual(L, R, bias, res, scale):
    """res + scale * (L [B, N, k] @ R [B, k, m] + bias) with autograd; call only when ``wide_matmul_residual_supported``."""
    return _WideMatmulResidual.apply(L, R, bias, res, scale)


class _TallReduceNT(torch.autograd.Function):
    """E [B, N, a]^T @ V [B, N, b] -> [B, a, b] (both operands token-major)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, E, V):
        ctx.save_for_backward(E, V)
        return xty(E, V)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        E, V = ctx.saved_tensors
        g = g.to(E.dtype)
        gE = _bmm_nt(V, g) if ctx.needs_input_grad[0] else None
        gV = _bmm_nn(E, g) if ctx.needs_input_grad[1] else None
        return gE, gV


def tall_reduce_nt(E, V):
    """E [B, N, a]^T @ V [B, N, b] with autograd."""
    return _TallReduceNT.apply(E, V)


# ---------------------------------------------------------------------------------------------
# Fused score post-processing of the bi-directional attention (csrc/bisoftmax.hip)
# ---------------------------------------------------------------------------------------------
_BIS_WS = {}


def _bis_workspace(lib, dev, B, N, H, T):
    n = lib.zira_bisoftmax_workspace_floats(B, N, H, T)
    return _scratch(_BIS_WS, dev, n)


def bi_softmax_supported(xm, H, T, dropout_active):
    return (xm.is_cuda and xm.dtype == torch.float32 and not dropout_active and H * T <= 2048
            and not torch.is_autocast_enabled())


def _bis_fwd(xm, c, ml, mv, B, N, H, T, stable, clo, chi):
    """csrc/bisoftmax.hip forward on contiguous fp32 tensors (masks as uint8 or None) -> pv, e, colsum, colmax, gmax"""
    from . import _lib

    lib = _lib.load()
    pv, e = torch.empty_like(xm), torch.empty_like(xm)
    colsum, colmax = torch.empty_like(c), torch.empty_like(c)
    gmax = torch.empty(1, dtype=torch.float32, device=xm.device)
    ws = _bis_workspace(lib, xm.device, B, N, H, T)
    rc = lib.zira_bisoftmax_fwd_f32(
        xm.data_ptr(), c.data_ptr(), ml.data_ptr() if ml is not None else None,
        mv.data_ptr() if mv is not None else None, B, N, H, T, int(stable), int(clo), int(chi),
        pv.data_ptr(), e.data_ptr(), colsum.data_ptr(), colmax.data_ptr(), gmax.data_ptr(), ws.data_ptr(),
        torch.cuda.current_stream(xm.device).cuda_stream)
    if rc != 0:
        raise RuntimeError("zira_bisoftmax_fwd_f32 failed with HIP error %d" % rc)
    return pv, e, colsum, colmax, gmax


def _bis_bwd(xm, c, ml, pv, e, colmax, gmax, g_pv, g_e, g_cs, B, N, H, T, stable, clo, chi):
    """csrc/bisoftmax.hip backward -> g_xm, g_c (None gradients count as zeros)"""
    from . import _lib

    lib = _lib.load()
    g_pv = torch.zeros_like(xm) if g_pv is None else g_pv.contiguous()
    g_e = torch.zeros_like(xm) if g_e is None else g_e.contiguous()
    g_cs = torch.zeros_like(c) if g_cs is None else g_cs.contiguous()
    g_xm, g_c = torch.empty_like(xm), torch.empty_like(c)
    ws = _bis_workspace(lib, xm.device, B, N, H, T)
    rc = lib.zira_bisoftmax_bwd_f32(
        xm.data_ptr(), c.data_ptr(), ml.data_ptr() if ml is not None else None, B, N, H, T,
        stable, clo, chi, pv.data_ptr(), e.data_ptr(), colmax.data_ptr(), gmax.data_ptr(), g_pv.data_ptr(),
        g_e.data_ptr(), g_cs.data_ptr(), g_xm.data_ptr(), g_c.data_ptr(), ws.data_ptr(),
        torch.cuda.current_stream(xm.device).cuda_stream)
    if rc != 0:
        raise RuntimeError("zira_bisoftmax_bwd_f32 failed with HIP error %d" % rc)
    return g_xm, g_c


class _BiSoftmax(torch.autograd.Function):
    """(xm [B,N,H*T], c [B,H*T], mask_l [B,T] | None, mask_v [B,N] | None) -> (pv, e, colsum); see
    include/zira_msda.h.  The gradient is only valid when e and colsum are used as e / colsum."""

    @staticmethod
    def forward(ctx, xm, c, mask_l, mask_v, H, T, stable, clamp_lo, clamp_hi):
        xm, c = xm.contiguous(), c.contiguous()
        B, N, HT = xm.shape
        assert HT == H * T and c.shape == (B, HT)
        ml = mask_l.contiguous().view(torch.uint8) if mask_l is not None else None
        mv = mask_v.contiguous().view(torch.uint8) if mask_v is not None else None
        pv, e, colsum, colmax, gmax = _bis_fwd(xm, c, ml, mv, B, N, H, T, stable, clamp_lo, clamp_hi)
        ctx.save_for_backward(xm, c, pv, e, colmax, gmax)
        ctx.ml = ml
        ctx.dims = (B, N, H, T, int(stable), int(clamp_lo), int(clamp_hi))
        return pv, e, colsum

    @staticmethod
    def backward(ctx, g_pv, g_e, g_cs):
        xm, c, pv, e, colmax, gmax = ctx.saved_tensors
        g_xm, g_c = _bis_bwd(xm, c, ctx.ml, pv, e, colmax, gmax, g_pv, g_e, g_cs, *ctx.dims)
        return g_xm, g_c, None, None, None, None, None, None, None


def bi_softmax(xm, c, mask_l, mask_v, H, T, stable=True, clamp_lo=True, clamp_hi=True):
    return _BiSoftmax.apply(xm, c, mask_l, mask_v, H, T, stable, clamp_lo, clamp_hi)


# ---- the image side of a fusion block as ONE autograd node ---------------------------------------
class _FusionImageSide(torch.autograd.Function):
    """(vn [B,N,C], a [B,C,HT], c [B,HT], z [B,HT,C], bias [C], scale [C] | [B,1,C]) ->
           out = vn + scale * (pv z + bias)   [B,N,C],    t = e^T vn   [B,HT,C],    colsum [B,HT]
    with (pv, e, colsum) = bi_softmax(vn a, c): ``wide_matmul`` -> ``bi_softmax`` -> ``tall_reduce_nt`` /
    ``wide_matmul_residual`` of BiAttentionBlock._forward_native_text (reference fuse_modules.py:170-248, :292-303 re-bracketed)
    held by one node, so that the backward forms the gradient of ``vn`` -- which autograd would add up from three nodes with
    two 45 MB passes -- in ONE pass over it:  g_vn = g_out + e g_t + g_xm a^T  is a contraction over the concatenated index
    with a residual (csrc/thin_f16x2.hip, two sources).  ``scale`` is constant (frozen layer scale x stochastic depth): it is
    folded into z and the bias as in ``_WideMatmulResidual``."""

    @staticmethod
    def forward(ctx, vn, a, c, z, bias, scale, mask_l, mask_v, H, T, stable, clamp_lo, clamp_hi):
        vn, a, c = vn.contiguous(), a.contiguous(), c.contiguous()
        B, N, C = vn.shape
        ml = mask_l.contiguous().view(torch.uint8) if mask_l is not None else None
        mv = mask_v.contiguous().view(torch.uint8) if mask_v is not None else None
        sc = scale if scale.dim() == 3 else scale.view(1, 1, -1)
        Rs = (z * sc).contiguous()                                                             # [B, HT, C]
        bs = (bias.view(1, -1) * sc.reshape(-1, sc.shape[-1])).expand(B, -1).contiguous()      # [B, C]
        xm = thin_bmm(vn, a, True)
        pv, e, colsum, colmax, gmax = _bis_fwd(xm, c, ml, mv, B, N, H, T, stable, clamp_lo, clamp_hi)
        t = xty(e, vn)
        out = thin_bmm(pv, Rs, True, bias=bs, res=vn)
        ctx.save_for_backward(vn, a, Rs, sc, xm, c, pv, e, colmax, gmax)
        ctx.ml = ml
        ctx.dims = (B, N, H, T, int(stable), int(clamp_lo), int(clamp_hi))
        return out, t, colsum

    @staticmethod
    @once_differentiable
    def backward(ctx, g_out, g_t, g_cs):
        vn, a, Rs, sc, xm, c, pv, e, colmax, gmax = ctx.saved_tensors
        B, N = ctx.dims[:2]
        g_out = torch.zeros_like(vn) if g_out is None else g_out.contiguous()
        g_t = torch.zeros_like(Rs) if g_t is None else g_t.contiguous()
        g_pv = thin_bmm(g_out, Rs, False)                                  # [B, N, HT]
        g_e = thin_bmm(vn, g_t, False)                                     # [B, N, HT]
        g_xm, g_c = _bis_bwd(xm, c, ctx.ml, pv, e, colmax, gmax, g_pv, g_e, g_cs, *ctx.dims)
        g_vn = thin_bmm(e, g_t, True, A2=g_xm, W2=a, w2_is_kn=False, res=g_out) if ctx.needs_input_grad[0] else None
        g_a = xty(vn, g_xm) if ctx.needs_input_grad[1] else None           # [B, C, HT]
        g_z = xty(pv, g_out) * sc if ctx.needs_input_grad[3] else None     # [B, HT, C]
        g_b = (g_out * sc).sum((0, 1)) if ctx.needs_input_grad[4] else None
        return g_vn, g_a, (g_c if ctx.needs_input_grad[2] else None), g_z, g_b, None, None, None, None, None, None, None, None


def fusion_image_side_supported(vn, a, z, bias, scale, H, T, dropout_active) -> bool:
    if not (vn.is_cuda and vn.dim() == 3 and a.dim() == 3 and z.dim() == 3 and bias is not None and bias.dim() == 1):
        return False
    if any(t.dtype != torch.float32 for t in (vn, a, z, bias, scale)) or scale.requires_grad or vn.shape[0] > 4:
        return False
    n, C = H * T, vn.shape[2]
    return (n <= 128 and a.shape == (vn.shape[0], C, n) and z.shape == (vn.shape[0], n, C) and thin_supported(vn, n, C)
            and thin_supported(vn, C, n, vn) and bi_softmax_supported(vn, H, T, dropout_active))


def fusion_image_side(vn, a, c, z, bias, scale, mask_l, mask_v, H, T, stable=True, clamp_lo=True, clamp_hi=True):
    """-> (vn + scale * (pv z + bias), e^T vn, colsum); call only when ``fusion_image_side_supported``."""
    return _FusionImageSide.apply(vn, a, c, z, bias, scale, mask_l, mask_v, H, T, stable, clamp_lo, clamp_hi)


# ---- GroupNorm of the input projections (csrc/groupnorm.hip) ---------------------------------------
_GN_WS = {}
USE_GROUP_NORM = True    # developer switch (scripts/ab_step.py group_norm=0): ATen's native_group_norm


def group_norm_supported(x, gn, res=None) -> bool:
    """NCHW fp32 on the GPU through a FROZEN nn.GroupNorm (the kernels return the input gradient only)."""
    if not (USE_GROUP_NORM and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and not torch.is_autocast_enabled("cuda")):
        return False
    if gn.weight is None or gn.bias is None or gn.weight.requires_grad or gn.bias.requires_grad or gn.weight.dtype != torch.float32:
        return False
    if res is not None and (res.shape != x.shape or res.dtype != torch.float32):
        return False
    B, C, H, W = x.shape
    from . import _lib
    return C == gn.num_channels and _lib.load().zira_groupnorm_workspace_floats(B, C, H * W, gn.num_groups) > 0


class _GroupNormFrozen(torch.autograd.Function):
    """y = GroupNorm(x (+ res)) with frozen affine parameters; the gradient goes to x (and res) unchanged in form."""

    @staticmethod
    def forward(ctx, x, res, weight, bias, groups, eps):
        from . import _lib

        lib = _lib.load()
        x = x.contiguous()
        res = res.contiguous() if res is not None else None
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        xs = torch.empty_like(x) if res is not None else x
        mean = torch.empty(B * groups, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        ws = _scratch(_GN_WS, x.device, lib.zira_groupnorm_workspace_floats(B, C, H * W, groups))
        with torch.cuda.device(x.device):
            rc = lib.zira_groupnorm_fwd_f32(x.data_ptr(), res.data_ptr() if res is not None else None, weight.data_ptr(), bias.data_ptr(),
                                            B, C, H * W, groups, float(eps), xs.data_ptr() if res is not None else None, y.data_ptr(),
                                            mean.data_ptr(), rstd.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("zira_groupnorm_fwd_f32 failed with code %d" % rc)
        ctx.save_for_backward(xs, weight, mean, rstd)
        ctx.groups, ctx.two = groups, res is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        from . import _lib

        lib = _lib.load()
        xs, weight, mean, rstd = ctx.saved_tensors
        gy = gy.contiguous()
        B, C, H, W = xs.shape
        dx = torch.empty_like(xs)
        ws = _scratch(_GN_WS, xs.device, lib.zira_groupnorm_workspace_floats(B, C, H * W, ctx.groups))
        with torch.cuda.device(xs.device):
            rc = lib.zira_groupnorm_bwd_f32(gy.data_ptr(), xs.data_ptr(), weight.data_ptr(), mean.data_ptr(), rstd.data_ptr(), B, C, H * W,
                                            ctx.groups, dx.data_ptr(), ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError("zira_groupnorm_bwd_f32 failed with code %d" % rc)
        return dx, (dx if ctx.two else None), None, None, None, None


def group_norm_frozen(x, gn, res=None):
    """``gn(x + res)`` (``gn(x)`` without res) for a frozen nn.GroupNorm; call when ``group_norm_supported``."""
    return _GroupNormFrozen.apply(x, res, gn.weight, gn.bias, gn.num_groups, gn.eps)


# ---- row LayerNorm (csrc/layernorm.hip) ------------------------------------------------------
LN_MIN_ROWS = 8192   # below this the launch is latency-bound either way and ATen's call path is leaner


def layer_norm_supported(x, normalized_shape, weight, bias):
    C = x.shape[-1] if x.dim() else 0
    return (x.is_cuda and x.dtype == torch.float32 and len(normalized_shape) == 1 and normalized_shape[0] == C
            and C % 4 == 0 and 0 < C <= 1024 and x.numel() // max(C, 1) >= LN_MIN_ROWS
            and (weight is None or weight.dtype == torch.float32) and (bias is None or bias.dtype == torch.float32))


class _LayerNorm(torch.autograd.Function):
    """Forward on the row kernel (x read once, 5-6 TB/s).  Backward: the input gradient on the matching row
    kernel when the affine parameters are frozen (the ZiRa fine-tune), otherwise
    aten::native_layer_norm_backward on the saved mean / rstd, i.e. what autograd of F.layer_norm runs."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, weight, bias, eps):
        from . import _lib
        C = x.shape[-1]
        xc = x.contiguous()
        rows = xc.numel() // C
        y = torch.empty_like(xc)
        stats = torch.empty((2, rows), device=x.device, dtype=torch.float32)
        w = weight.contiguous() if weight is not None else None
        b = bias.contiguous() if bias is not None else None
        rc = _lib.load().zira_layernorm_fwd_f32(
            xc.data_ptr(), w.data_ptr() if w is not None else None, b.data_ptr() if b is not None else None,
            rows, C, float(eps), y.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(),
            torch.cuda.current_stream(x.device).cuda_stream)
        if rc != 0:
            raise RuntimeError("zira_layernorm_fwd_f32 failed with code %d" % rc)
        ctx.save_for_backward(xc, w, b, stats)
        ctx.C = C
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        xc, w, b, stats = ctx.saved_tensors
        lead = xc.shape[:-1] + (1,)
        mask = [ctx.needs_input_grad[0], w is not None and ctx.needs_input_grad[1],
                b is not None and ctx.needs_input_grad[2]]
        gy = gy.contiguous()
        if LayerNorm.fused_backward and mask[0] and not mask[1] and not mask[2] and gy.dtype == torch.float32:   # frozen affine: dx only
            from . import _lib
            gx = torch.empty_like(xc)
            rc = _lib.load().zira_layernorm_bwd_f32(
                gy.data_ptr(), xc.data_ptr(), w.data_ptr() if w is not None else None, stats[0].data_ptr(),
                stats[1].data_ptr(), xc.numel() // ctx.C, ctx.C, gx.data_ptr(),
                torch.cuda.current_stream(xc.device).cuda_stream)
            if rc != 0:
                raise RuntimeError("zira_layernorm_bwd_f32 failed with code %d" % rc)
            return gx, None, None, None
        gx, gw, gb = torch.ops.aten.native_layer_norm_backward(
            gy, xc, [ctx.C], stats[0].view(lead), stats[1].view(lead), w, b, mask)
        return gx, gw, gb, None


class _AddLayerNorm(torch.autograd.Function):
    """LayerNorm(x + res) in one pass (zira_add_layernorm_fwd_f32): the residual connection in front of a post-LN.
    The sum is formed with the rounding of a separate add and saved for the backward, whose input gradient goes to
    x and res alike."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, x, res, weight, bias, eps):
        from . import _lib
        C = x.shape[-1]
        xc, rc_ = x.contiguous(), res.contiguous()
        rows = xc.numel() // C
        y, s = torch.empty_like(xc), torch.empty_like(xc)
        stats = torch.empty((2, rows), device=x.device, dtype=torch.float32)
        w = weight.contiguous() if weight is not None else None
        b = bias.contiguous() if bias is not None else None
        rc = _lib.load().zira_add_layernorm_fwd_f32(
            xc.data_ptr(), rc_.data_ptr(), w.data_ptr() if w is not None else None,
            b.data_ptr() if b is not None else None, rows, C, float(eps), s.data_ptr(), y.data_ptr(),
            stats[0].data_ptr(), stats[1].data_ptr(), torch.cuda.current_stream(x.device).cuda_stream)
        if rc != 0:
            raise RuntimeError("zira_add_layernorm_fwd_f32 failed with code %d" % rc)
        ctx.save_for_backward(s, w, b, stats)
        ctx.C = C
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, gy):
        s, w, b, stats = ctx.saved_tensors
        lead = s.shape[:-1] + (1,)
        need_in = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        mask = [need_in, w is not None and ctx.needs_input_grad[2], b is not None and ctx.needs_input_grad[3]]
        gy = gy.contiguous()
        if LayerNorm.fused_backward and mask[0] and not mask[1] and not mask[2] and gy.dtype == torch.float32:
            from . import _lib
            gs = torch.empty_like(s)
            rc = _lib.load().zira_layernorm_bwd_f32(
                gy.data_ptr(), s.data_ptr(), w.data_ptr() if w is not None else None, stats[0].data_ptr(),
                stats[1].data_ptr(), s.numel() // ctx.C, ctx.C, gs.data_ptr(),
                torch.cuda.current_stream(s.device).cuda_stream)
            if rc != 0:
                raise RuntimeError("zira_layernorm_bwd_f32 failed with code %d" % rc)
            gw = gb = None
        else:
            gs, gw, gb = torch.ops.aten.native_layer_norm_backward(
                gy, s, [ctx.C], stats[0].view(lead), stats[1].view(lead), w, b, mask)
        return (gs if ctx.needs_input_grad[0] else None, gs if ctx.needs_input_grad[1] else None, gw, gb, None)


def add_layer_norm(x, res, normalized_shape, weight=None, bias=None, eps=1e-5):
    """F.layer_norm(x + res, ...); one kernel where layer_norm() would use the row kernel."""
    normalized_shape = tuple(normalized_shape) if not isinstance(normalized_shape, int) else (normalized_shape,)
    if (LayerNorm.fused_residual and x.shape == res.shape and x.dtype == res.dtype
            and layer_norm_supported(x, normalized_shape, weight, bias)):
        return _AddLayerNorm.apply(x, res, weight, bias, eps)
    return layer_norm(x + res, normalized_shape, weight, bias, eps)


def layer_norm(x, normalized_shape, weight=None, bias=None, eps=1e-5):
    """F.layer_norm with the forward of wide fp32 activations on csrc/layernorm.hip."""
    normalized_shape = tuple(normalized_shape) if not isinstance(normalized_shape, int) else (normalized_shape,)
    if layer_norm_supported(x, normalized_shape, weight, bias):
        return _LayerNorm.apply(x, weight, bias, eps)
    return F.layer_norm(x, normalized_shape, weight, bias, eps)


class LayerNorm(torch.nn.LayerNorm):
    """nn.LayerNorm (same parameters and state-dict keys) whose forward goes through layer_norm()."""

    fused = True            # class-level switches for A/B runs
    fused_backward = True
    fused_residual = True

    def forward(self, x):
        if self.fused:
            return layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
        return super().forward(x)

    def add_norm(self, x, res):
        """self(x + res)"""
        if self.fused:
            return add_layer_norm(x, res, self.normalized_shape, self.weight, self.bias, self.eps)
        return super().forward(x + res)
