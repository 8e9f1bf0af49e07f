"""The host-memory twins of the C ABI (zira_msda_{fwd,bwd}_cpu_f32, csrc/msda_cpu.cpp: product code, not the
oracle) against the golden vectors produced by the reference's own function (tests/golden/gen_msda_golden.py).
No GPU involved.  fp32 fixtures only (the twins are float32); tolerance 2e-5 of the tensor scale."""
import ctypes
import os

import numpy as np
import pytest

from conftest import golden_msda_cases, load_npz
from test_oracle_golden import _on_minus_one_edge as _minus_one_edge
from ziragroundingdino_amd import _lib

F32 = [p for p in golden_msda_cases() if load_npz(p)["value"].dtype == np.float32]


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize("path", F32, ids=lambda p: os.path.basename(p)[5:-4])
def test_cpu_abi_matches_reference_golden(path):
    lib = _lib.load()
    g = load_npz(path)
    value, loc, attn, go = (np.ascontiguousarray(g[k]) for k in ("value", "sampling_loc", "attn_weight", "grad_output"))
    sh = np.ascontiguousarray(g["spatial_shapes"], dtype=np.int64)
    st = np.ascontiguousarray(g["level_start_index"], dtype=np.int64)
    B, S, M, D = value.shape
    Q, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
    out = np.full((B, Q, M * D), np.nan, np.float32)
    assert lib.zira_msda_fwd_cpu_f32(_ptr(value), _ptr(sh), _ptr(st), _ptr(loc), _ptr(attn), B, S, M, D, L, Q, P, _ptr(out)) == 0
    gv, gl, ga = (np.full(a.shape, np.nan, np.float32) for a in (value, loc, attn))   # (every element is written)
    assert lib.zira_msda_bwd_cpu_f32(_ptr(go), _ptr(value), _ptr(sh), _ptr(st), _ptr(loc), _ptr(attn), B, S, M, D, L, Q, P,
                                     _ptr(gv), _ptr(gl), _ptr(ga)) == 0

    def close(got, want, what):
        scale = max(1.0, float(np.abs(want).max()))
        err = float(np.abs(got - want).max()) / scale
        assert err <= 2e-5, "%s: max err %.3e (scaled)" % (what, err)

    close(out, g["output"], "output")
    close(gv, g["grad_value"], "grad_value")
    close(ga, g["grad_attn_weight"], "grad_attn_weight")
    want_gl = g["grad_sampling_loc"].copy()
    edge = _minus_one_edge(g)   # pixel coordinate exactly -1: the kernel semantics (cuh:288) give zero there
    assert not gl[edge].any()
    want_gl[edge] = 0
    close(gl, want_gl, "grad_sampling_loc")


def test_cpu_abi_rejects_bad_arguments():
    lib = _lib.load()
    assert lib.zira_msda_fwd_cpu_f32(None, None, None, None, None, 1, 1, 1, 32, 1, 1, 1, None) == 1
    assert lib.zira_msda_bwd_cpu_f32(None, None, None, None, None, None, 1, 1, 1, 32, 1, 1, 1, None, None, None) == 1
