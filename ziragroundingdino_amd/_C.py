"""Drop-in for the reference's native module ``groundingdino._C``.

Same two functions, same argument order and meaning, same error behaviour as the pybind
module built from reference csrc/vision.cpp:53-56 (implementation
csrc/MsDeformAttn/ms_deform_attn.h:21-61 and ms_deform_attn_cuda.cu:21-154), but backed by the
hand-written gfx950 kernels behind the C ABI of include/zira_msda.h.

    from ziragroundingdino_amd import _C
    out = _C.ms_deform_attn_forward(value, spatial_shapes, level_start_index,
                                    sampling_loc, attn_weight, im2col_step)
    gv, gl, ga = _C.ms_deform_attn_backward(value, spatial_shapes, level_start_index,
                                            sampling_loc, attn_weight, grad_output, im2col_step)
"""
import torch

from . import _lib

_SUFFIX = {torch.float32: "f32", torch.float64: "f64"}


def _check_common(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, extra=()):
    named = (("value", value), ("spatial_shapes", spatial_shapes),
             ("level_start_index", level_start_index), ("sampling_loc", sampling_loc),
             ("attn_weight", attn_weight)) + tuple(extra)
    # ms_deform_attn.h:29-40 -- device dispatch happens on `value` first
    if not value.is_cuda:
        raise RuntimeError("Not implemented on the CPU")
    # ms_deform_attn_cuda.cu:29-39 / :94-106
    for name, t in named:
        if not t.is_contiguous():
            raise RuntimeError("%s tensor has to be contiguous" % name)
    for name, t in named:
        if not t.is_cuda:
            raise RuntimeError("%s must be a CUDA tensor" % name)
    if value.dtype not in _SUFFIX:
        # AT_DISPATCH_FLOATING_TYPES (ms_deform_attn_cuda.cu:65)
        raise RuntimeError('"ms_deform_attn_forward_cuda" not implemented for \'%s\''
                           % str(value.dtype).replace("torch.", "").capitalize())
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("expected scalar type Long for spatial_shapes / level_start_index")
    if sampling_loc.dtype != value.dtype or attn_weight.dtype != value.dtype:
        raise RuntimeError("expected sampling_loc and attn_weight to have the dtype of value")
    if value.dim() != 4 or sampling_loc.dim() != 6 or attn_weight.dim() != 5 \
            or spatial_shapes.dim() != 2:
        raise RuntimeError("ms_deform_attn: value[B,S,M,D], spatial_shapes[L,2], "
                           "sampling_loc[B,Q,M,L,P,2], attn_weight[B,Q,M,L,P] expected")


def _dims(value, spatial_shapes, sampling_loc, im2col_step):
    B, S, M, D = value.shape
    L = spatial_shapes.shape[0]
    Q = sampling_loc.shape[1]
    P = sampling_loc.shape[4]
    step = min(B, int(im2col_step))
    # ms_deform_attn_cuda.cu:51-53
    if step <= 0 or B % step != 0:
        raise RuntimeError("batch(%d) must divide im2col_step(%d)" % (B, step))
    return B, S, M, D, L, Q, P


USE_TILED_BACKWARD = True  # False forces the atomic backward (tests compare the two)
USE_FORWARD_PLAN = True    # False: sparse backward calls plan inside the backward call (zira_msda_bwd_f32_ws)
# Optional launch timing for bench.py: when TIMING is a list, every call appends
# (kind, (B, S, M, D, L, Q, P), start_event, end_event) recorded on the launch stream.
TIMING = None


class _Timed:
    def __init__(self, kind, dims):
        self.kind, self.dims = kind, dims

    def __enter__(self):
        if TIMING is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if TIMING is not None:
            self.e1.record()
            TIMING.append((self.kind, self.dims, self.e0, self.e1))
        return False
_WS_CACHE = {}


def _workspace_bytes(lib, *dims):
    n = _WS_CACHE.get(dims)
    if n is None:
        n = _WS_CACHE[dims] = int(lib.zira_msda_bwd_workspace_bytes(*dims))
    return n


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _raise_on(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: hipError %d" % (what, rc))


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                           im2col_step):
    """-> output[B, Q, M*D].  Reference: ms_deform_attn_cuda_forward (ms_deform_attn_cuda.cu:21-81)."""
    _check_common(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    B, S, M, D, L, Q, P = _dims(value, spatial_shapes, sampling_loc, im2col_step)
    lib = _lib.load()
    out = torch.empty((B, Q, M * D), dtype=value.dtype, device=value.device)
    fn = getattr(lib, "zira_msda_fwd_" + _SUFFIX[value.dtype])
    args = (value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
            sampling_loc.data_ptr(), attn_weight.data_ptr(), B, S, M, D, L, Q, P, out.data_ptr())
    with torch.cuda.device(value.device), _Timed("fwd", (B, S, M, D, L, Q, P)):
        rc = fn(*args, _stream())
    _raise_on(rc, "ms_deform_attn_forward")
    return out


_PLAN_CACHE = {}


def _plan_bytes(lib, *dims):
    n = _PLAN_CACHE.get(dims)
    if n is None:
        n = _PLAN_CACHE[dims] = int(lib.zira_msda_plan_bytes(*dims))
    return n


def _plan_dims(value, spatial_shapes, level_start_index, sampling_loc, im2col_step):
    """Dimensions and plan size of a call the planned backward serves, else None."""
    if not (USE_FORWARD_PLAN and USE_TILED_BACKWARD and value.is_cuda and sampling_loc.is_cuda
            and value.dtype == torch.float32 and sampling_loc.dtype == torch.float32 and sampling_loc.is_contiguous()
            and spatial_shapes.is_cuda and level_start_index.is_cuda and value.dim() == 4 and sampling_loc.dim() == 6):
        return None
    dims = _dims(value, spatial_shapes, sampling_loc, im2col_step)
    nbytes = _plan_bytes(_lib.load(), *dims)
    return (dims, nbytes) if nbytes else None


def _plan_key(sampling_loc, attn_weight):
    """What a plan was made from: it holds the tile of every sample and the attention weight of every record, so it serves the
    backward of exactly these two tensors in exactly this state."""
    return (sampling_loc.data_ptr(), sampling_loc._version, tuple(sampling_loc.shape),
            attn_weight.data_ptr(), attn_weight._version)


def plan_applies(value, spatial_shapes, level_start_index, sampling_loc, im2col_step):
    """True where ``ms_deform_attn_forward_plan`` / ``ms_deform_attn_plan`` return a plan: sparse float32 calls on the
    GPU (decoder cross-attention) with D = 32; False for CPU tensors, dense calls, other dtypes / widths."""
    return _plan_dims(value, spatial_shapes, level_start_index, sampling_loc, im2col_step) is not None


def ms_deform_attn_plan(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    """-> plan (an opaque uint8 device tensor) or None.  The backward of a sparse float32 call (decoder
    cross-attention) is cut into tiles by a plan that depends on the level tables, the sampling locations and (for the
    records' weights) the attention weights only, so
    it can be made in the forward pass, off the backward's critical path: this enqueues the planning kernel on the
    current stream (C ABI ``zira_msda_plan_f32``); hand the result to ``ms_deform_attn_backward(..., plan=plan)``.
    None where no planned backward exists (``plan_applies``) or with ``USE_FORWARD_PLAN`` off: the backward then plans
    by itself."""
    pd = _plan_dims(value, spatial_shapes, level_start_index, sampling_loc, im2col_step)
    if pd is None:
        return None
    dims, nbytes = pd
    lib = _lib.load()
    plan = torch.empty(nbytes, dtype=torch.uint8, device=value.device)
    with torch.cuda.device(value.device), _Timed("plan", dims):
        rc = lib.zira_msda_plan_f32(spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
                                    attn_weight.data_ptr(), *dims, plan.data_ptr(), nbytes, _stream())
    _raise_on(rc, "ms_deform_attn_plan")
    plan.zira_plan_key = _plan_key(sampling_loc, attn_weight)
    return plan


def ms_deform_attn_forward_plan(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    """-> (output[B, Q, M*D], plan): ``ms_deform_attn_forward`` and ``ms_deform_attn_plan`` in ONE launch (C ABI
    ``zira_msda_fwd_plan_f32``: the plan's blocks run beside the gather's, the only way the two overlap on this stack).
    Only where ``plan_applies``."""
    _check_common(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    pd = _plan_dims(value, spatial_shapes, level_start_index, sampling_loc, im2col_step)
    if pd is None:
        raise RuntimeError("ms_deform_attn_forward_plan: no planned backward for this call (see plan_applies)")
    dims, nbytes = pd
    B, S, M, D, L, Q, P = dims
    lib = _lib.load()
    out = torch.empty((B, Q, M * D), dtype=value.dtype, device=value.device)
    plan = torch.empty(nbytes, dtype=torch.uint8, device=value.device)
    with torch.cuda.device(value.device), _Timed("fwd", dims):
        rc = lib.zira_msda_fwd_plan_f32(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                                        sampling_loc.data_ptr(), attn_weight.data_ptr(), *dims, out.data_ptr(),
                                        plan.data_ptr(), nbytes, _stream())
    _raise_on(rc, "ms_deform_attn_forward_plan")
    plan.zira_plan_key = _plan_key(sampling_loc, attn_weight)
    return out, plan


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                            grad_output, im2col_step, plan=None):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight].
    Reference: ms_deform_attn_cuda_backward (ms_deform_attn_cuda.cu:84-154).  ``plan``: the second result of
    ``ms_deform_attn_forward_planned`` for the same level tables and sampling locations (optional)."""
    _check_common(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                  extra=(("grad_output", grad_output),))
    B, S, M, D, L, Q, P = _dims(value, spatial_shapes, sampling_loc, im2col_step)
    if grad_output.dtype != value.dtype:
        raise RuntimeError("expected grad_output to have the dtype of value")
    lib = _lib.load()
    grad_value = torch.empty_like(value)          # written exactly once by the callee
    grad_loc = torch.empty_like(sampling_loc)
    grad_attn = torch.empty_like(attn_weight)
    args = (grad_output.data_ptr(), value.data_ptr(), spatial_shapes.data_ptr(),
            level_start_index.data_ptr(), sampling_loc.data_ptr(), attn_weight.data_ptr(),
            B, S, M, D, L, Q, P, grad_value.data_ptr(), grad_loc.data_ptr(), grad_attn.data_ptr())
    if plan is not None:
        if value.dtype != torch.float32 or plan.dtype != torch.uint8 or not plan.is_cuda \
                or plan.numel() < _plan_bytes(lib, B, S, M, D, L, Q, P) or not _plan_bytes(lib, B, S, M, D, L, Q, P):
            raise RuntimeError("ms_deform_attn_backward: plan does not belong to this call")
        key = getattr(plan, "zira_plan_key", None)   # (a plan made through this module; a raw buffer filled through the C ABI has none)
        if key is not None and key != _plan_key(sampling_loc, attn_weight):
            raise RuntimeError("ms_deform_attn_backward: the plan was made for other sampling locations / attention weights "
                               "(or they were modified in place since): its tiles and record weights do not fit this call")
        with torch.cuda.device(value.device), _Timed("bwd", (B, S, M, D, L, Q, P)):
            rc = lib.zira_msda_bwd_planned_f32(*args, plan.data_ptr(), plan.numel(), _stream())
        _raise_on(rc, "ms_deform_attn_backward")
        return [grad_value, grad_loc, grad_attn]
    ws_bytes = _workspace_bytes(lib, B, S, M, D, L, Q, P) if value.dtype == torch.float32 else 0
    tiled = bool(ws_bytes) and USE_TILED_BACKWARD
    if tiled:  # atomic-free two-kernel backward; scratch comes from torch's caching allocator
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=value.device)
    with torch.cuda.device(value.device), _Timed("bwd", (B, S, M, D, L, Q, P)):
        if tiled:
            rc = lib.zira_msda_bwd_f32_ws(*args, ws.data_ptr(), ws_bytes, _stream())
        else:
            rc = getattr(lib, "zira_msda_bwd_" + _SUFFIX[value.dtype])(*args, _stream())
    _raise_on(rc, "ms_deform_attn_backward")
    return [grad_value, grad_loc, grad_attn]
