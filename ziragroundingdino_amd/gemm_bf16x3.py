"""Host side of ``zira_gemm_bf16x3_f32`` / ``zira_split_bf16x3_f32`` (csrc/gemm_bf16x3.hip): fp32-accurate products of the
image-token rows with FROZEN weights on the bf16 matrix cores -- each fp32 operand is exactly the sum of three bfloat16
numbers, six exact product terms reproduce the fp32 product to 2^-26, the sums are fp32 inside the matrix core.  The
weight's three planes are made once and follow the parameter in place (captured graphs keep reading the same buffers);
the activation is split inside the kernel.  Stands for ``F.linear`` / autograd's ``mm`` in the reference FFN
(transformer_for_adapter.py:877-886) under the freeze of groundingdino_dual_zero_rep_branch.py:722-745.

No autograd here: the callers are hand-written forward / backward pairs (transformer._FrozenFFN, _FrozenFFNNorm)."""
import torch

from . import _lib

EPI_BIAS, EPI_BIAS_RELU, EPI_MASK, EPI_ADD = 0, 1, 2, 3


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def split_planes(weight: torch.Tensor, transpose: bool, out: torch.Tensor = None) -> torch.Tensor:
    """weight [rows, cols] fp32 on the GPU -> int16 [3, N, K] holding the bfloat16 planes, B[n][k] = weight[n][k]
    (``transpose`` False) or weight[k][n] (True)."""
    assert weight.is_cuda and weight.dtype == torch.float32 and weight.dim() == 2 and weight.is_contiguous()
    rows, cols = weight.shape
    N, K = (cols, rows) if transpose else (rows, cols)
    if out is None:
        out = torch.empty((3, N, K), device=weight.device, dtype=torch.int16)
    assert out.shape == (3, N, K) and out.dtype == torch.int16 and out.is_contiguous()
    with torch.cuda.device(weight.device):
        rc = _lib.load().zira_split_bf16x3_f32(weight.data_ptr(), rows, cols, 1 if transpose else 0, out.data_ptr(), _stream(weight))
    if rc != 0:
        raise RuntimeError("zira_split_bf16x3_f32 failed with code %d" % rc)
    return out


class SplitWeight:
    """The bf16 planes of one frozen weight in one orientation, refreshed IN PLACE when the parameter changes
    (``data_ptr`` / ``_version``): a replayed hipGraph keeps reading the same buffer."""

    def __init__(self, transpose: bool):
        self.transpose, self.key, self.buf = transpose, None, None

    def planes(self, weight: torch.Tensor) -> torch.Tensor:
        key = (weight.data_ptr(), weight._version, weight.device)
        if key != self.key:
            with torch.no_grad():
                same = self.buf is not None and self.buf.device == weight.device and self.buf.numel() == 3 * weight.numel()
                self.buf = split_planes(weight.detach(), self.transpose, self.buf if same else None)
            self.key = key
        return self.buf


def supported(a: torch.Tensor, N: int, K: int) -> bool:
    return (a.is_cuda and a.dtype == torch.float32 and a.dim() == 2 and a.is_contiguous() and a.shape[1] == K
            and N % 128 == 0 and K % 32 == 0 and a.data_ptr() % 16 == 0)


def gemm(a: torch.Tensor, planes: torch.Tensor, epilogue: int, bias: torch.Tensor = None, aux: torch.Tensor = None,
         out: torch.Tensor = None) -> torch.Tensor:
    """epilogue(a [M, K] @ B^T) -> [M, N] with B = ``planes`` [3, N, K] (split_planes).  ``out`` may be ``aux`` (EPI_ADD)."""
    M, K = a.shape
    N = planes.shape[1]
    assert planes.shape == (3, N, K) and planes.dtype == torch.int16 and planes.is_contiguous() and planes.device == a.device
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    assert out.shape == (M, N) and out.is_contiguous() and out.dtype == torch.float32
    if bias is not None:
        assert bias.shape == (N,) and bias.is_contiguous() and bias.dtype == torch.float32
    if aux is not None:
        assert aux.shape == (M, N) and aux.is_contiguous() and aux.dtype == torch.float32
    with torch.cuda.device(a.device):
        rc = _lib.load().zira_gemm_bf16x3_f32(a.data_ptr(), planes.data_ptr(), M, N, K, epilogue,
                                              0 if bias is None else bias.data_ptr(), 0 if aux is None else aux.data_ptr(),
                                              out.data_ptr(), _stream(a))
    if rc != 0:
        raise RuntimeError("zira_gemm_bf16x3_f32 failed with code %d (M=%d N=%d K=%d epilogue=%d)" % (rc, M, N, K, epilogue))
    return out


# ---- cached split weights of modules -----------------------------------------------------------------------------------------

def enabled() -> bool:
    """Whether the callers should take the split-bf16 products (``transformer.Switches.gemm_arith`` = "bf16x3", or "f16x2":
    the fused f16x2 FFN launches with these products everywhere else)."""
    from .transformer import Switches
    return Switches.gemm_arith in ("bf16x3", "f16x2")


def _cache(owner, name, transpose):
    store = owner.__dict__.setdefault("_bf16x3_split", {})
    sw = store.get((name, transpose))
    if sw is None:
        sw = store[(name, transpose)] = SplitWeight(transpose)
    return sw


def linear(owner, name, x2, weight, bias=None, out=None):
    """``x2 @ weight.T (+ bias)`` for a frozen ``weight`` [N, K]; the planes are cached on ``owner`` under ``name``."""
    N = weight.shape[0]
    planes = _cache(owner, name, False).planes(weight)
    if bias is None:
        bias = _zeros(N, x2.device)
    return gemm(x2, planes, EPI_BIAS, bias=bias, out=out)


def linear_input_grad(owner, name, g2, weight, accumulate_into=None):
    """``g2 @ weight`` for a frozen ``weight`` [N_out, K_in] (the input gradient of ``F.linear``): [M, N_out] -> [M, K_in];
    ``accumulate_into`` [M, K_in]: added to IN PLACE (the gradient that meets this one) and returned."""
    planes = _cache(owner, name, True).planes(weight)          # B[n][k] = weight[k][n]
    if accumulate_into is not None:
        return gemm(g2, planes, EPI_ADD, aux=accumulate_into, out=accumulate_into)
    return gemm(g2, planes, EPI_BIAS, bias=_zeros(weight.shape[1], g2.device))


_ZEROS = {}


def _zeros(n, device):
    z = _ZEROS.get((n, device))
    if z is None:
        z = _ZEROS[(n, device)] = torch.zeros(n, device=device, dtype=torch.float32)
    return z


def refresh(owner, weights):
    """Bring every cached plane set of ``owner`` up to date (``weights``: name -> tensor), in place; for
    ``refresh_fused_projection`` hooks (a replayed hipGraph re-runs no Python)."""
    for (name, _), sw in owner.__dict__.get("_bf16x3_split", {}).items():
        w = weights.get(name)
        if w is not None:
            sw.planes(w)
