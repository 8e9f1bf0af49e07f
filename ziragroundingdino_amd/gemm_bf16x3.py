"""Host side of ``zira_gemm_bf16x3_f32`` / ``zira_split_bf16x3_f32`` (csrc/gemm_bf16x3.hip): fp32-accurate products of the
image-token rows with FROZEN weights on the bf16 matrix cores -- each fp32 operand is exactly the sum of three bfloat16
numbers, six exact product terms reproduce the fp32 product to 2^-26, the sums are fp32 inside the matrix core.  The
weight's three planes are made once and follow the parameter in place (captured graphs keep reading the same buffers);
the activation is split inside the kernel.  Stands for ``F.linear`` / autograd's ``mm`` in the reference FFN
(transformer_for_adapter.py:877-886) under the freeze of groundingdino_dual_zero_rep_branch.py:722-745.

Under ``transformer.Switches.gemm_arith = "f16x2"`` the cached helpers at the end of this file (``linear``,
``linear_input_grad``, ``refresh``) take the two-plane f16 form of the same products instead (csrc/gemm_f16x2.hip: three
terms per fragment pair instead of six; ``split_planes_f16x2`` / ``gemm_f16x2`` below).

No autograd here: the callers are hand-written forward / backward pairs (transformer._FrozenFFN, _FrozenFFNNorm)."""
import torch

from . import _lib

EPI_BIAS, EPI_BIAS_RELU, EPI_MASK, EPI_ADD = 0, 1, 2, 3
EPI_BIAS_GELU, EPI_BIAS_RES = 4, 5      # the tiled f16x2 kernel only (csrc/gemm_f16x2.hip)
CALLS = {"gemm_f16x2": 0}              # launches so far (tests)


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


def f16x2_plane_halves(N: int, K: int) -> int:
    """int16 elements of split_planes_f16x2's result: two planes [N][K rounded up to 32] and N floats"""
    return 2 * N * ((K + 31) // 32 * 32) + 2 * N


def split_planes_f16x2(weight: torch.Tensor, transpose: bool, out: torch.Tensor = None) -> torch.Tensor:
    """weight [rows, cols] fp32 on the GPU -> flat int16 [2 N K + 2 N]: the two f16 planes [2, N, K] of the row-scaled weight,
    then 1 / scale of every row as N floats; B[n][k] = weight[n][k] (``transpose`` False) or weight[k][n] (True)."""
    assert weight.is_cuda and weight.dtype == torch.float32 and weight.dim() == 2 and weight.is_contiguous()
    rows, cols = weight.shape
    N, K = (cols, rows) if transpose else (rows, cols)
    n = f16x2_plane_halves(N, K)
    if out is None:
        out = torch.empty(n, device=weight.device, dtype=torch.int16)
    assert out.shape == (n,) and out.dtype == torch.int16 and out.is_contiguous()
    with torch.cuda.device(weight.device):
        rc = _lib.load().zira_split_f16x2_f32(weight.data_ptr(), rows, cols, 1 if transpose else 0, out.data_ptr(), _stream(weight))
    if rc != 0:
        raise RuntimeError("zira_split_f16x2_f32 failed with code %d" % rc)
    return out


def split_frags_f16x2(weight: torch.Tensor, transpose: bool, out: torch.Tensor = None) -> torch.Tensor:
    """As ``split_planes_f16x2`` with the halves in the order the matrix core reads them ([N / 32][K / 16][plane][lane][8]):
    the operand of ``gemm_f16x2_panel`` (N % 32 == 0, K % 16 == 0)."""
    assert weight.is_cuda and weight.dtype == torch.float32 and weight.dim() == 2 and weight.is_contiguous()
    rows, cols = weight.shape
    n = 2 * rows * cols + 2 * (cols if transpose else rows)
    if out is None:
        out = torch.empty(n, device=weight.device, dtype=torch.int16)
    assert out.shape == (n,) and out.dtype == torch.int16 and out.is_contiguous()
    with torch.cuda.device(weight.device):
        rc = _lib.load().zira_split_f16x2_frag_f32(weight.data_ptr(), rows, cols, 1 if transpose else 0, out.data_ptr(), _stream(weight))
    if rc != 0:
        raise RuntimeError("zira_split_f16x2_frag_f32 failed with code %d" % rc)
    return out


USE_PANEL = True   # module-level switch for A/B runs (scripts/ab_step.py panel=0|1)


def panel_supported(N: int, K: int) -> bool:
    """The skinny products csrc/gemm_f16x2_panel.hip takes: all of K in one block's LDS panel, a few column tiles per wave --
    the 256-wide projections of the deformable attention and their input gradients (the backbone's K = 384 linears have
    8400 rows and up to 1536 columns: 263 blocks walking 48 column tiles each lose to the tiled kernel, 53 against 30-80 us)."""
    return USE_PANEL and N % 32 == 0 and ((K == 256 and N <= 512) or (K == 384 and N == 256))


class SplitWeight:
    """The planes of one frozen weight in one orientation (bf16 x 3; with ``f16x2`` f16 x 2, in fragment order where the
    panel kernel takes the product), refreshed IN PLACE when the parameter changes (``data_ptr`` / ``_version``): a replayed
    hipGraph keeps reading the same buffer."""

    def __init__(self, transpose: bool, f16x2: bool = False, panel: bool = False):
        self.transpose, self.key, self.buf, self.f16x2, self.panel = transpose, None, None, f16x2, panel
        self.shape = None

    def planes(self, weight: torch.Tensor) -> torch.Tensor:
        key = (weight.data_ptr(), weight._version, weight.device)
        if key != self.key:
            with torch.no_grad():
                same = self.buf is not None and self.buf.device == weight.device and self.shape == tuple(weight.shape)
                fn = (split_frags_f16x2 if self.panel else split_planes_f16x2) if self.f16x2 else split_planes
                self.buf = fn(weight.detach(), self.transpose, self.buf if same else None)
                self.shape = tuple(weight.shape)
            self.key = key
        return self.buf


def supported(a: torch.Tensor, N: int, K: int) -> bool:
    return (a.is_cuda and a.dtype == torch.float32 and a.dim() == 2 and a.is_contiguous() and a.shape[1] == K
            and N % 128 == 0 and K % 32 == 0 and a.data_ptr() % 16 == 0)


def supported_f16x2(a: torch.Tensor, N: int, K: int) -> bool:
    """The tiled two-plane f16 kernel alone: N a multiple of 32 (``supported`` is what all three arithmetics take)."""
    return (_two_plane() and a.is_cuda and a.dtype == torch.float32 and a.dim() == 2 and a.is_contiguous() and a.shape[1] == K
            and N % 32 == 0 and K % 32 == 0 and a.data_ptr() % 16 == 0)


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


def gemm_f16x2(a: torch.Tensor, planes: torch.Tensor, N: int, epilogue: int, bias: torch.Tensor = None, aux: torch.Tensor = None,
               out: torch.Tensor = None, row_scale: torch.Tensor = None, rows_per_scale: int = 0) -> torch.Tensor:
    """epilogue(a [M, K] @ B^T) -> [M, N] with B = ``planes`` (split_planes_f16x2 of an [N, K] weight).  ``out`` may be ``aux``.
    ``row_scale`` [ceil(M / rows_per_scale)] with EPI_BIAS_RES: aux + row_scale[m // rows_per_scale] * (product + bias)."""
    M, K = a.shape
    assert planes.shape == (f16x2_plane_halves(N, K),) and planes.dtype == torch.int16 and planes.is_contiguous() and planes.device == a.device
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    assert out.shape == (M, N) and out.is_contiguous() and out.dtype == torch.float32
    if bias is not None:
        assert bias.shape == (N,) and bias.is_contiguous() and bias.dtype == torch.float32
    if aux is not None:
        assert aux.shape == (M, N) and aux.is_contiguous() and aux.dtype == torch.float32
    if row_scale is not None:
        assert row_scale.dtype == torch.float32 and row_scale.is_contiguous() and rows_per_scale > 0
        assert row_scale.numel() * rows_per_scale >= M
    with torch.cuda.device(a.device):
        rc = _lib.load().zira_gemm_f16x2_ex_f32(a.data_ptr(), planes.data_ptr(), M, N, K, epilogue,
                                                0 if bias is None else bias.data_ptr(), 0 if aux is None else aux.data_ptr(),
                                                0 if row_scale is None else row_scale.data_ptr(), int(rows_per_scale),
                                                out.data_ptr(), _stream(a))
    if rc != 0:
        raise RuntimeError("zira_gemm_f16x2_ex_f32 failed with code %d (M=%d N=%d K=%d epilogue=%d)" % (rc, M, N, K, epilogue))
    CALLS["gemm_f16x2"] += 1
    return out


def gemm_f16x2_panel(a: torch.Tensor, frags: torch.Tensor, N: int, epilogue: int, bias: torch.Tensor = None, aux: torch.Tensor = None,
                     out: torch.Tensor = None, add: torch.Tensor = None) -> torch.Tensor:
    """epilogue((a [+ add]) [M, K] @ B^T) -> [M, N] with B = ``frags`` (split_frags_f16x2 of an [N, K] weight), K = 256 or 384."""
    M, K = a.shape
    assert panel_supported(N, K) and frags.shape == (2 * N * K + 2 * N,) and frags.dtype == torch.int16 and frags.device == a.device
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    assert out.shape == (M, N) and out.is_contiguous() and out.dtype == torch.float32
    if bias is not None:
        assert bias.shape == (N,) and bias.is_contiguous() and bias.dtype == torch.float32
    if aux is not None:
        assert aux.shape == (M, N) and aux.is_contiguous() and aux.dtype == torch.float32
    if add is not None:
        assert add.shape == a.shape and add.is_contiguous() and add.dtype == torch.float32 and add.data_ptr() % 16 == 0
    with torch.cuda.device(a.device):
        rc = _lib.load().zira_gemm_f16x2_panel_f32(a.data_ptr(), 0 if add is None else add.data_ptr(), frags.data_ptr(), M, N, K, epilogue,
                                                   0 if bias is None else bias.data_ptr(), 0 if aux is None else aux.data_ptr(),
                                                   out.data_ptr(), _stream(a))
    if rc != 0:
        raise RuntimeError("zira_gemm_f16x2_panel_f32 failed with code %d (M=%d N=%d K=%d epilogue=%d)" % (rc, M, N, K, epilogue))
    return out


# ---- cached split weights of modules -----------------------------------------------------------------------------------------

def _two_plane() -> bool:
    from .transformer import Switches
    return Switches.gemm_arith == "f16x2"


def enabled() -> bool:
    """Whether the callers should take the split-bf16 products (``transformer.Switches.gemm_arith`` = "bf16x3", or "f16x2":
    the fused f16x2 FFN launches with these products everywhere else)."""
    from .transformer import Switches
    return Switches.gemm_arith in ("bf16x3", "f16x2")


def _cache(owner, name, transpose, N, K):
    # (one store per owner; the arithmetics and kernels keep their planes under different keys)
    store = owner.__dict__.setdefault("_bf16x3_split", {})
    two = _two_plane()
    panel = two and panel_supported(N, K)
    key = (name, transpose, "panel" if panel else "tiled") if two else (name, transpose)
    sw = store.get(key)
    if sw is None:
        sw = store[key] = SplitWeight(transpose, f16x2=two, panel=panel)
    return sw


def _gemm_cached(sw, a, weight, N, epilogue, add=None, **kw):
    planes = sw.planes(weight)
    if sw.panel:
        return gemm_f16x2_panel(a, planes, N, epilogue, add=add, **kw)
    if add is not None:
        a = a + add
    if sw.f16x2:
        return gemm_f16x2(a, planes, N, epilogue, **kw)
    return gemm(a, planes, epilogue, **kw)


def linear(owner, name, x2, weight, bias=None, out=None, add=None):
    """``(x2 [+ add]) @ weight.T (+ bias)`` for a frozen ``weight`` [N, K]; the planes are cached on ``owner`` under ``name``.
    ``add`` [M, K]: a second operand summed with ``x2`` on the way in (inside the panel kernel where it takes the product)."""
    N, K = weight.shape
    if bias is None:
        bias = _zeros(N, x2.device)
    return _gemm_cached(_cache(owner, name, False, N, K), x2, weight, N, EPI_BIAS, add=add, bias=bias, out=out)


def linear_tiled_f16x2(owner, name, x2, weight, bias, epilogue=EPI_BIAS, residual=None, row_scale=None, rows_per_scale=0, out=None):
    """``epilogue(x2 @ weight.T + bias)`` through the TILED two-plane f16 kernel whatever the shape (the backbone's linears:
    N % 32 == 0): EPI_BIAS, EPI_BIAS_GELU, or EPI_BIAS_RES with ``residual`` [M, N] and an optional ``row_scale`` (one factor
    per ``rows_per_scale`` rows).  Planes cached on ``owner`` as in ``linear``."""
    N, K = weight.shape
    store = owner.__dict__.setdefault("_bf16x3_split", {})
    key = (name, False, "tiled")
    sw = store.get(key)
    if sw is None:
        sw = store[key] = SplitWeight(False, f16x2=True, panel=False)
    return gemm_f16x2(x2, sw.planes(weight), N, epilogue, bias=bias, aux=residual, out=out, row_scale=row_scale,
                      rows_per_scale=rows_per_scale)


def linear_input_grad(owner, name, g2, weight, accumulate_into=None):
    """``g2 @ weight`` for a frozen ``weight`` [N_out, K_in] (the input gradient of ``F.linear``): [M, N_out] -> [M, K_in];
    ``accumulate_into`` [M, K_in]: added to IN PLACE (the gradient that meets this one) and returned."""
    K, N = weight.shape                                             # B[n][k] = weight[k][n]
    sw = _cache(owner, name, True, N, K)
    if accumulate_into is not None:
        return _gemm_cached(sw, g2, weight, N, EPI_ADD, aux=accumulate_into, out=accumulate_into)
    return _gemm_cached(sw, g2, weight, N, EPI_BIAS, bias=_zeros(N, g2.device))


_ZEROS = {}


def _zeros(n, device):
    z = _ZEROS.get((n, device))
    if z is None:
        z = _ZEROS[(n, device)] = torch.zeros(n, device=device, dtype=torch.float32)
    return z


def refresh(owner, weights):
    """Bring every cached plane set of ``owner`` up to date (``weights``: name -> tensor), in place; for
    ``refresh_fused_projection`` hooks (a replayed hipGraph re-runs no Python)."""
    for key, sw in owner.__dict__.get("_bf16x3_split", {}).items():
        w = weights.get(key[0])
        if w is not None:
            sw.planes(w)
