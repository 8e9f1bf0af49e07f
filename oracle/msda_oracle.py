"""numpy/ctypes front-end of oracle/msda_oracle.c -- TEST INFRASTRUCTURE ONLY.

The arithmetic lives in the C file (which cites the reference lines it restates); this
module only builds/loads ``oracle/_build/libmsda_oracle.so`` and marshals numpy arrays.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmsda_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (a few hundred ms)."""
    src = os.path.join(_HERE, "msda_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.msda_oracle_num_threads.restype = ctypes.c_int
    return _lib


def num_threads() -> int:
    return int(_load().msda_oracle_num_threads())


def set_num_threads(n: int) -> None:
    _load().msda_oracle_set_num_threads(ctypes.c_int(int(n)))


def _suffix(dtype):
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise TypeError("oracle supports float32/float64 only, got %s" % dtype)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _prep(value, shapes, start, loc, attn):
    dt = value.dtype
    value = np.ascontiguousarray(value)
    loc = np.ascontiguousarray(loc, dtype=dt)
    attn = np.ascontiguousarray(attn, dtype=dt)
    shapes = np.ascontiguousarray(shapes, dtype=np.int64)
    start = np.ascontiguousarray(start, dtype=np.int64)
    B, S, M, D = value.shape
    _, Q, M2, L, P, two = loc.shape
    assert M2 == M and two == 2 and shapes.shape == (L, 2) and start.shape == (L,)
    assert attn.shape == (B, Q, M, L, P)
    assert int((shapes[:, 0] * shapes[:, 1]).sum()) == S
    return value, shapes, start, loc, attn, (B, S, M, D, L, Q, P)


def msda_forward(value, shapes, start, loc, attn):
    """out[B,Q,M*D] for value[B,S,M,D], loc[B,Q,M,L,P,2], attn[B,Q,M,L,P] (numpy)."""
    value, shapes, start, loc, attn, (B, S, M, D, L, Q, P) = _prep(value, shapes, start, loc, attn)
    out = np.empty((B, Q, M * D), dtype=value.dtype)
    fn = getattr(_load(), "msda_oracle_fwd_" + _suffix(value.dtype))
    rc = fn(_p(value), _p(shapes), _p(start), _p(loc), _p(attn),
            B, S, M, D, L, Q, P, _p(out))
    assert rc == 0
    return out


def msda_backward(grad_out, value, shapes, start, loc, attn):
    """(grad_value, grad_loc, grad_attn) -- same layouts as the inputs."""
    value, shapes, start, loc, attn, (B, S, M, D, L, Q, P) = _prep(value, shapes, start, loc, attn)
    grad_out = np.ascontiguousarray(grad_out, dtype=value.dtype)
    assert grad_out.shape == (B, Q, M * D)
    gv = np.empty_like(value)
    gl = np.empty_like(loc)
    ga = np.empty_like(attn)
    fn = getattr(_load(), "msda_oracle_bwd_" + _suffix(value.dtype))
    rc = fn(_p(grad_out), _p(value), _p(shapes), _p(start), _p(loc), _p(attn),
            B, S, M, D, L, Q, P, _p(gv), _p(gl), _p(ga))
    assert rc == 0
    return gv, gl, ga
