"""Parity of the HIP kernels (through the `_C` drop-in -> C ABI) against
(1) the golden vectors produced by the reference's own CPU path and
(2) the CPU oracle on seeded inputs, up to the BASELINE shapes; plus size-independent
properties at full size and the reference's error behaviour at the boundary.

Tolerances (north_star: "within 1e-3 fp32"): we hold fp32 results to 2e-5 relative to the
tensor's scale -- the only differences are summation order (4 corners x 16 samples folded by
shuffles, atomics in grad_value) -- and fp64 to 1e-11.
"""
import os
import zlib

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_msda_cases, load_npz

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from ziragroundingdino_amd import _C, MultiScaleDeformableAttnFunction

DEV = "cuda"
NORTH_STAR_SHAPES = [(100, 167), (50, 84), (25, 42), (13, 21)]


def _tol(dtype):
    return 2e-5 if dtype in (np.float32, torch.float32) else 1e-11


def _close(got, want, tol, what):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else got
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got - want).max()) / scale
    assert err <= tol, "%s: max err %.3e (scaled) > %.1e" % (what, err, tol)


def _to_dev(g):
    t = lambda k: torch.from_numpy(g[k]).to(DEV)
    return (t("value"), t("spatial_shapes"), t("level_start_index"), t("sampling_loc"),
            t("attn_weight"), t("grad_output"))


def _minus_one_edge(g):
    sh = g["spatial_shapes"]
    dt = g["value"].dtype
    W = sh[:, 1][None, None, None, :, None].astype(dt)
    H = sh[:, 0][None, None, None, :, None].astype(dt)
    loc = g["sampling_loc"]
    return (loc[..., 0] * W - 0.5 == -1) | (loc[..., 1] * H - 0.5 == -1)


@pytest.mark.parametrize("path", golden_msda_cases(), ids=lambda p: os.path.basename(p)[5:-4])
def test_hip_matches_reference_golden(path):
    g = load_npz(path)
    tol = _tol(g["value"].dtype.type)
    value, shapes, start, loc, attn, go = _to_dev(g)
    out = _C.ms_deform_attn_forward(value, shapes, start, loc, attn, 64)
    _close(out, g["output"], tol, "output")
    gv, gl, ga = _C.ms_deform_attn_backward(value, shapes, start, loc, attn, go, 64)
    _close(gv, g["grad_value"], tol, "grad_value")
    _close(ga, g["grad_attn_weight"], tol, "grad_attn_weight")
    ref_gl = g["grad_sampling_loc"].copy()
    edge = _minus_one_edge(g)  # see tests/test_oracle_golden.py: CUDA-kernel semantics there
    gl = gl.cpu().numpy()
    assert not gl[edge].any()
    ref_gl[edge] = 0
    _close(gl, ref_gl, tol, "grad_sampling_loc")


def _random_case(B, Q, M, D, shapes, P, seed, dtype=np.float32, lo=-0.1, hi=1.1, clustered=False, hot=False,
                 inmodel=False):
    rng = np.random.default_rng(seed)
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    value = rng.standard_normal((B, S, M, D)).astype(dtype)
    if hot:  # every query looks at one of three spots: a few tiles receive thousands of entries
        spots = rng.uniform(0.2, 0.8, (3, 2))
        centre = spots[rng.integers(0, 3, (B, Q))][:, :, None, None, None, :]
        # hot == 2: exactly AT the spots -- a dozen grad_value rows per head and level take all entries
        loc = (centre + (0.01 if hot == 1 else 0.0) * rng.standard_normal((B, Q, M, L, P, 2))).astype(dtype)
    elif clustered:  # decoder-like: box centre + small offsets
        centre = rng.uniform(0.1, 0.9, (B, Q, 1, 1, 1, 2))
        loc = (centre + 0.05 * rng.standard_normal((B, Q, M, L, P, 2))).astype(dtype)
    else:
        loc = rng.uniform(lo, hi, (B, Q, M, L, P, 2)).astype(dtype)
    logits = rng.standard_normal((B, Q, M, L * P))
    attn = np.exp(logits - logits.max(-1, keepdims=True))
    attn = (attn / attn.sum(-1, keepdims=True)).reshape(B, Q, M, L, P).astype(dtype)
    if inmodel:  # sampling locations / attention weights captured from a training step of the full-size model
        # at random init (last decoder layer, scripts/inmodel_msda.py): 8 % of the backward's tiles are heavy
        with np.load(os.path.join(GOLDEN, "inmodel_decoder_locations.npz")) as z:
            loc, attn = z["loc"].astype(dtype), z["attn"].astype(dtype)
        assert loc.shape == (B, Q, M, L, P, 2)
    go = rng.standard_normal((B, Q, M * D)).astype(dtype)
    sh = np.asarray(shapes, dtype=np.int64)
    start = np.concatenate([[0], np.cumsum(sh[:, 0] * sh[:, 1])[:-1]]).astype(np.int64)
    return value, sh, start, loc, attn, go


CASES = [
    # (id, B, Q, M, D, shapes, P, kwargs)
    ("northstar_decoder", 2, 900, 8, 32, NORTH_STAR_SHAPES, 4, dict(lo=0.0, hi=1.0)),
    ("northstar_clustered", 2, 900, 8, 32, NORTH_STAR_SHAPES, 4, dict(clustered=True)),
    ("northstar_hot_tiles", 2, 900, 8, 32, NORTH_STAR_SHAPES, 4, dict(hot=True)),
    ("northstar_pinpoint", 2, 900, 8, 32, NORTH_STAR_SHAPES, 4, dict(hot=2)),
    ("northstar_inmodel", 2, 900, 8, 32, NORTH_STAR_SHAPES, 4, dict(inmodel=True)),
    ("d16_hot_tiles", 2, 700, 8, 16, [(40, 61), (20, 31)], 4, dict(hot=True)),
    ("d64_pinpoint", 1, 700, 4, 64, [(40, 61), (20, 31)], 4, dict(hot=2)),
    ("d64_hot_tiles", 1, 700, 4, 64, [(40, 61), (20, 31)], 4, dict(hot=True)),
    ("oob_heavy", 2, 333, 8, 32, [(20, 31), (10, 16), (5, 8), (3, 4)], 4, dict(lo=-0.5, hi=1.5)),
    ("lp_not_16", 3, 57, 4, 32, [(12, 9), (6, 5), (3, 3), (2, 2), (1, 1)], 5, {}),
    ("one_level_p1", 2, 200, 2, 32, [(17, 23)], 1, {}),
    ("d16", 2, 101, 8, 16, [(20, 31), (10, 16)], 4, {}),
    ("d64", 2, 101, 4, 64, [(20, 31), (10, 16)], 4, {}),
    ("d128", 1, 64, 2, 128, [(9, 7), (4, 4)], 2, {}),
    ("d4", 1, 77, 3, 4, [(9, 7), (4, 4)], 4, {}),
    ("d24_generic", 2, 50, 3, 24, [(9, 7), (4, 4)], 4, {}),
    ("h1_w1_levels", 2, 90, 4, 32, [(1, 9), (7, 1), (1, 1)], 4, dict(lo=-0.3, hi=1.3)),
    ("f64", 2, 150, 8, 32, [(20, 31), (10, 16), (5, 8), (3, 4)], 4, dict(dtype=np.float64)),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_hip_matches_oracle(oracle, case):
    _, B, Q, M, D, shapes, P, kw = case
    value, sh, start, loc, attn, go = _random_case(B, Q, M, D, shapes, P, seed=zlib.crc32(case[0].encode()) % 1000, **kw)
    tol = _tol(value.dtype.type)
    want_out = oracle.msda_forward(value, sh, start, loc, attn)
    want_gv, want_gl, want_ga = oracle.msda_backward(go, value, sh, start, loc, attn)
    t = lambda a: torch.from_numpy(a).to(DEV)
    tv, tsh, tst, tloc, tattn, tgo = map(t, (value, sh, start, loc, attn, go))
    out = _C.ms_deform_attn_forward(tv, tsh, tst, tloc, tattn, 64)
    _close(out, want_out, tol, "output")
    gv, gl, ga = _C.ms_deform_attn_backward(tv, tsh, tst, tloc, tattn, tgo, 64)
    _close(gv, want_gv, tol, "grad_value")
    _close(ga, want_ga, tol, "grad_attn_weight")
    # the kernels form the pixel coordinate exactly like the oracle (mul, then sub, no fma),
    # so floor() agrees everywhere and grad_loc can be compared on every sample
    _close(gl, want_gl, tol, "grad_sampling_loc")


@pytest.mark.parametrize("pattern", ["gauss2px", "init_grid"])
def test_encoder_shape_against_oracle(oracle, pattern):
    """Q = S (every pixel is a query, reference transformer_for_adapter.py:893-900) at the
    benchmarked batch size B = 2 (the grid and the per-XCD split differ from B = 1).  Two location
    patterns: pixel-grid reference points + N(0, 2 px) offsets, and the module's own
    initial offsets -- head m points along direction m, point p at (p + 1) pixels, identical for
    every query (reference ms_deform_attn.py:194-217) -- which is what a freshly built model feeds the
    op and is as regular as inputs get (every sample of a head / point lands on the same sub-pixel
    phase; whole rows of queries hit the same cells)."""
    shapes = NORTH_STAR_SHAPES
    sh = np.asarray(shapes, dtype=np.int64)
    S = int((sh[:, 0] * sh[:, 1]).sum())
    B, M, D, L, P = 2, 8, 32, 4, 4
    rng = np.random.default_rng(5)
    value, _, start, _, attn, go = _random_case(B, S, M, D, shapes, P, seed=5)
    # reference points = pixel centres of every level's grid, replicated over levels
    ref = np.concatenate([
        np.stack(np.meshgrid((np.arange(w) + 0.5) / w, (np.arange(h) + 0.5) / h), -1).reshape(-1, 2)
        for h, w in shapes]).astype(np.float32)                                  # [S,2] (x,y)
    if pattern == "gauss2px":
        off_px = 2.0 * rng.standard_normal((B, S, M, L, P, 2)).astype(np.float32)
    else:
        theta = np.arange(M, dtype=np.float32) * (2.0 * np.pi / M)
        d = np.stack([np.cos(theta), np.sin(theta)], -1)
        d = d / np.abs(d).max(-1, keepdims=True)
        off_px = (d[None, None, :, None, None, :] * np.arange(1, P + 1, dtype=np.float32)[None, None, None, None, :, None])
        off_px = np.broadcast_to(off_px, (B, S, M, L, P, 2)).astype(np.float32)
        attn = np.full_like(attn, 1.0 / (L * P))
    norm = np.stack([sh[:, 1], sh[:, 0]], -1).astype(np.float32)[None, None, None, :, None, :]
    loc = np.ascontiguousarray((ref[None, :, None, None, None, :] + off_px / norm).astype(np.float32))
    attn = np.ascontiguousarray(attn)
    want_out = oracle.msda_forward(value, sh, start, loc, attn)
    want = oracle.msda_backward(go, value, sh, start, loc, attn)
    t = lambda a: torch.from_numpy(a).to(DEV)
    tv, tsh, tst, tloc, tattn, tgo = map(t, (value, sh, start, loc, attn, go))
    _close(_C.ms_deform_attn_forward(tv, tsh, tst, tloc, tattn, 64), want_out, 2e-5, "output")
    got = _C.ms_deform_attn_backward(tv, tsh, tst, tloc, tattn, tgo, 64)
    for g, w, name in zip(got, want, ("grad_value", "grad_loc", "grad_attn")):
        _close(g, w, 5e-5 if name == "grad_value" else 2e-5, name)


@pytest.mark.parametrize("D,M,P,shapes", [(16, 8, 4, [(64, 80), (32, 40), (16, 20)]), (64, 4, 2, [(96, 120), (48, 60), (24, 30)]),
                                          (32, 8, 3, [(70, 91), (35, 46)])])
def test_dense_backward_other_widths_against_oracle(oracle, D, M, P, shapes):
    """The dense (cell walk) backward at the other specialised channel widths and level / point counts (the model only
    ever calls it with D = 32, L = 4, P = 4): queries on the pixel grid of every level, N(0, 1.5 px) offsets plus a few
    far-away and out-of-range samples."""
    sh = np.asarray(shapes, dtype=np.int64)
    S = int((sh[:, 0] * sh[:, 1]).sum())
    B, L = 2, len(shapes)
    assert B * M * S >= 16 * 4096, "not a dense call"
    rng = np.random.default_rng(D + P)
    value, _, start, _, attn, go = _random_case(B, S, M, D, shapes, P, seed=D)
    ref = np.concatenate([
        np.stack(np.meshgrid((np.arange(w) + 0.5) / w, (np.arange(h) + 0.5) / h), -1).reshape(-1, 2)
        for h, w in shapes]).astype(np.float32)
    off_px = 1.5 * rng.standard_normal((B, S, M, L, P, 2)).astype(np.float32)
    far = rng.random((B, S, M, L, P)) < 0.02
    off_px[far] += rng.uniform(-60, 60, (int(far.sum()), 2)).astype(np.float32)
    norm = np.stack([sh[:, 1], sh[:, 0]], -1).astype(np.float32)[None, None, None, :, None, :]
    loc = np.ascontiguousarray((ref[None, :, None, None, None, :] + off_px / norm).astype(np.float32))
    want_out = oracle.msda_forward(value, sh, start, loc, attn)
    want = oracle.msda_backward(go, value, sh, start, loc, attn)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    tv, tsh, tst, tloc, tattn, tgo = map(t, (value, sh, start, loc, attn, go))
    _close(_C.ms_deform_attn_forward(tv, tsh, tst, tloc, tattn, 64), want_out, 2e-5, "output")
    got = _C.ms_deform_attn_backward(tv, tsh, tst, tloc, tattn, tgo, 64)
    for g, w, name in zip(got, want, ("grad_value", "grad_loc", "grad_attn")):
        _close(g, w, 5e-5 if name == "grad_value" else 2e-5, name)


def _dense_case(seed, B=2, M=8, D=32, P=4, shapes=((64, 80), (32, 40), (16, 20))):
    """Queries on the pixel grid of every level (Q = S), N(0, 1.5 px) offsets: a dense call (heads * Q >= 65536)."""
    sh = np.asarray(shapes, dtype=np.int64)
    S = int((sh[:, 0] * sh[:, 1]).sum())
    L = len(shapes)
    assert B * M * S >= 16 * 4096
    rng = np.random.default_rng(seed)
    value, _, start, _, attn, go = _random_case(B, S, M, D, shapes, P, seed=seed)
    ref = np.concatenate([
        np.stack(np.meshgrid((np.arange(w) + 0.5) / w, (np.arange(h) + 0.5) / h), -1).reshape(-1, 2)
        for h, w in shapes]).astype(np.float32)
    off_px = 1.5 * rng.standard_normal((B, S, M, L, P, 2)).astype(np.float32)
    norm = np.stack([sh[:, 1], sh[:, 0]], -1).astype(np.float32)[None, None, None, :, None, :]
    loc = np.ascontiguousarray((ref[None, :, None, None, None, :] + off_px / norm).astype(np.float32))
    return value, sh, start, loc, attn, go


@pytest.mark.parametrize("gscale,ascale", [(1e-30, 1.0), (1e30, 1.0), (1.0, 37.5), (3e-12, 1e-3)])
def test_dense_backward_fixed_point_scale(oracle, gscale, ascale):
    """The dense D = 32 backward sums grad_value in 64-bit fixed point whose scale follows max|grad_out| * max|attn|:
    the result must not depend on the magnitude of either."""
    value, sh, start, loc, attn, go = _dense_case(5)
    go = (go * np.float32(gscale)).astype(np.float32)
    attn = (attn * np.float32(ascale)).astype(np.float32)
    want = oracle.msda_backward(go, value, sh, start, loc, attn)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    got = _C.ms_deform_attn_backward(*map(t, (value, sh, start, loc, attn, go)), 64)
    for g, w, name in zip(got, want, ("grad_value", "grad_loc", "grad_attn")):
        w = np.asarray(w, dtype=np.float64)
        unit = float(np.abs(w).max())
        assert unit > 0 and np.isfinite(unit)
        err = float(np.abs(g.cpu().numpy().astype(np.float64) - w).max()) / unit
        assert err <= 5e-5, "%s: %.3e of the largest entry" % (name, err)


@pytest.mark.parametrize("path", ["dense", "sparse"])
def test_backward_with_one_outsized_head(oracle, path):
    """Wide dynamic range across heads: the gradients of ONE (image, head) pair are 1e6 times the others'.  The dense
    D = 32 backward scales its fixed-point sums per head (a call-wide scale would leave the quiet heads ~2^-18 of the
    quantum's headroom: 1e-3 relative errors), the sparse one sums in double: every head's grad_value must be right
    RELATIVE TO ITS OWN magnitude."""
    if path == "dense":
        value, sh, start, loc, attn, go = _dense_case(7)
    else:
        value, sh, start, loc, attn, go = _random_case(2, 900, 8, 32, NORTH_STAR_SHAPES, 4, seed=7, lo=0.0, hi=1.0)
    B, S, M, D = value.shape
    go = go.reshape(B, -1, M, D).copy()
    go[1, :, 3, :] *= np.float32(1e6)
    go = go.reshape(B, -1, M * D)
    want = oracle.msda_backward(go, value, sh, start, loc, attn)[0].astype(np.float64)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    got = _C.ms_deform_attn_backward(*map(t, (value, sh, start, loc, attn, go)), 64)[0].cpu().numpy().astype(np.float64)
    for b in range(B):
        for m in range(M):
            w, g = want[b, :, m, :], got[b, :, m, :]
            unit = float(np.abs(w).max())
            assert unit > 0
            err = float(np.abs(g - w).max()) / unit
            assert err <= 5e-5, "head (%d, %d): %.3e of its own largest entry" % (b, m, err)


def test_sparse_backward_time_does_not_depend_on_clustering():
    """The decoder-shape backward must cost about the same wherever the queries look: the locations captured from
    a training step (objects: a few tiles take most samples), all queries on three spots, all queries AT three
    pixels -- each within 1.3x of uniform locations (round 2's entry sort took 1.6-1.7x there).  hipGraph replays of
    10 launches, best of 5."""
    def timed(case_kw):
        value, sh, start, loc, attn, go = _random_case(2, 900, 8, 32, NORTH_STAR_SHAPES, 4, seed=11, **case_kw)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
        args = list(map(t, (value, sh, start, loc, attn, go)))
        fn = lambda: _C.ms_deform_attn_backward(*args, 64)
        for _ in range(3):
            fn()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10):
                fn()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 100.0)   # us per launch
        return best
    uniform = timed(dict(lo=0.0, hi=1.0))
    for name, kw in (("inmodel", dict(inmodel=True)), ("hot", dict(hot=True)), ("pinpoint", dict(hot=2))):
        us = timed(kw)
        assert us <= 1.3 * uniform, "%s: %.1f us against %.1f us on uniform locations" % (name, us, uniform)


def test_dense_backward_is_run_to_run_identical():
    value, sh, start, loc, attn, go = _dense_case(6)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    args = list(map(t, (value, sh, start, loc, attn, go)))
    first = _C.ms_deform_attn_backward(*args, 64)
    for _ in range(3):
        again = _C.ms_deform_attn_backward(*args, 64)
        for a, b in zip(first, again):
            assert torch.equal(a, b)


@pytest.mark.parametrize("poison", [float("inf"), float("nan")])
def test_dense_backward_non_finite_grad_out(oracle, poison):
    """Non-finite gradients (an overflowed loss scale) cannot be summed in fixed point: the call falls back to the
    float path and propagates them like the reference does."""
    value, sh, start, loc, attn, go = _dense_case(7)
    go.reshape(go.shape[0], go.shape[1], -1)[1, 1234, 3 * 32 + 5] = poison
    want = oracle.msda_backward(go, value, sh, start, loc, attn)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    got = _C.ms_deform_attn_backward(*map(t, (value, sh, start, loc, attn, go)), 64)
    for g, w, name in zip(got, want, ("grad_value", "grad_loc", "grad_attn")):
        g = g.cpu().numpy()
        fin = np.isfinite(w)
        assert not fin.all(), name if name != "grad_value" else "oracle did not propagate"
        assert not np.isfinite(g[~fin]).any(), "%s: finite where the reference is not" % name
        scale = max(1.0, float(np.abs(w[fin]).max()))
        assert float(np.abs(g[fin] - w[fin]).max()) / scale <= 5e-5, name


def test_full_size_properties():
    """BASELINE shape (B=2,Q=900,M=8,D=32,L=4,P=4), no oracle: linearity in value / attn /
    grad_out, zero for fully-outside samples, adjointness <out, go> == <value, grad_value>."""
    value, sh, start, loc, attn, go = _random_case(2, 900, 8, 32, NORTH_STAR_SHAPES, 4, seed=11)
    t = lambda a: torch.from_numpy(a).to(DEV)
    v, tsh, tst, tloc, tattn, tgo = map(t, (value, sh, start, loc, attn, go))
    v2 = torch.randn_like(v)
    f = lambda vv, aa: _C.ms_deform_attn_forward(vv, tsh, tst, tloc, aa, 64)
    o1, o2 = f(v, tattn), f(v2, tattn)
    torch.testing.assert_close(f(2 * v - 3 * v2, tattn), 2 * o1 - 3 * o2, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(f(v, 0.25 * tattn), 0.25 * o1, rtol=1e-5, atol=1e-6)
    gv, gl, ga = _C.ms_deform_attn_backward(v, tsh, tst, tloc, tattn, tgo, 64)
    # adjoint identity: the op is linear in value, so <f(v), go> == <v, grad_value>
    lhs = (o1.double() * tgo.double()).sum()
    rhs = (v.double() * gv.double()).sum()
    assert abs(lhs - rhs) <= 1e-5 * max(1.0, abs(lhs)), (lhs, rhs)
    # ... and linear in attn: <f, go> == <attn, grad_attn>
    rhs_a = (tattn.double() * ga.double()).sum()
    assert abs(lhs - rhs_a) <= 1e-5 * max(1.0, abs(lhs)), (lhs, rhs_a)
    gv2, gl2, ga2 = _C.ms_deform_attn_backward(v, tsh, tst, tloc, tattn, 2 * tgo, 64)
    torch.testing.assert_close(gl2, 2 * gl, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(gv2, 2 * gv, rtol=1e-4, atol=1e-5)
    far = torch.full_like(tloc, 2.5)
    assert not _C.ms_deform_attn_forward(v, tsh, tst, far, tattn, 64).any()
    z = _C.ms_deform_attn_backward(v, tsh, tst, far, tattn, tgo, 64)
    assert not z[0].any() and not z[1].any() and not z[2].any()


def test_outputs_need_no_preinit_and_rerun_is_stable():
    value, sh, start, loc, attn, go = _random_case(2, 64, 8, 32, [(8, 9), (4, 5)], 4, seed=3)
    t = lambda a: torch.from_numpy(a).to(DEV)
    args = list(map(t, (value, sh, start, loc, attn)))
    tgo = t(go)
    a = _C.ms_deform_attn_backward(*args, tgo, 64)
    b = _C.ms_deform_attn_backward(*args, tgo, 64)
    torch.testing.assert_close(a[0], b[0], rtol=1e-5, atol=1e-6)   # atomics: order may differ
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])      # shuffle reductions: bitwise
    assert torch.equal(_C.ms_deform_attn_forward(*args, 64), _C.ms_deform_attn_forward(*args, 64))


def test_autograd_function_and_stream():
    value, sh, start, loc, attn, go = _random_case(2, 40, 8, 32, [(8, 9), (4, 5)], 4, seed=4)
    t = lambda a: torch.from_numpy(a).to(DEV)
    v, tsh, tst, tloc, tattn, tgo = map(t, (value, sh, start, loc, attn, go))
    v.requires_grad_(True); tloc.requires_grad_(True); tattn.requires_grad_(True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):  # the op must launch on torch's *current* stream
        out = MultiScaleDeformableAttnFunction.apply(v, tsh, tst, tloc, tattn, 64)
        out.backward(tgo)
    torch.cuda.current_stream().wait_stream(side)
    ref = _C.ms_deform_attn_backward(v.detach(), tsh, tst, tloc.detach(), tattn.detach(), tgo, 64)
    torch.testing.assert_close(v.grad, ref[0], rtol=1e-5, atol=1e-6)
    assert torch.equal(tloc.grad, ref[1]) and torch.equal(tattn.grad, ref[2])


def test_boundary_errors_match_reference():
    value, sh, start, loc, attn, go = _random_case(3, 8, 2, 32, [(4, 5)], 2, seed=6)
    t = lambda a: torch.from_numpy(a).to(DEV)
    v, tsh, tst, tloc, tattn, tgo = map(t, (value, sh, start, loc, attn, go))
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        _C.ms_deform_attn_forward(v.cpu(), tsh, tst, tloc, tattn, 64)
    with pytest.raises(RuntimeError, match="spatial_shapes must be a CUDA tensor"):
        _C.ms_deform_attn_forward(v, tsh.cpu(), tst, tloc, tattn, 64)
    with pytest.raises(RuntimeError, match="value tensor has to be contiguous"):
        _C.ms_deform_attn_forward(v.transpose(2, 3), tsh, tst, tloc, tattn, 64)
    with pytest.raises(RuntimeError, match="must divide im2col_step"):
        _C.ms_deform_attn_forward(v, tsh, tst, tloc, tattn, 2)     # batch 3 % 2 != 0
    with pytest.raises(RuntimeError, match="grad_output tensor has to be contiguous"):
        _C.ms_deform_attn_backward(v, tsh, tst, tloc, tattn, tgo.transpose(0, 1).contiguous().transpose(0, 1), 64)
    with pytest.raises(RuntimeError, match="not implemented for"):
        _C.ms_deform_attn_forward(v.half(), tsh, tst, tloc.half(), tattn.half(), 64)


def test_tiled_and_atomic_backward_agree(oracle):
    """The default backward is the atomic-free two-kernel path (C ABI zira_msda_bwd_f32_ws);
    forcing the atomic path (zira_msda_bwd_f32) must give the same three gradients."""
    from ziragroundingdino_amd import _lib

    lib = _lib.load()
    for (B, Q, M, D, shapes, P) in [(2, 900, 8, 32, NORTH_STAR_SHAPES, 4),
                                    (1, 100, 4, 32, [(16, 20)], 4),          # BASELINE configs[0]
                                    (3, 131, 5, 16, [(9, 11), (4, 6), (2, 3)], 3),
                                    (2, 77, 2, 64, [(30, 41), (15, 21)], 8)]:  # LP = 16 / 9 / 16
        S = sum(h * w for h, w in shapes)
        assert lib.zira_msda_bwd_workspace_bytes(B, S, M, D, len(shapes), Q, P) > 0
        value, sh, start, loc, attn, go = _random_case(B, Q, M, D, shapes, P, seed=21, lo=-0.2, hi=1.2)
        t = lambda a: torch.from_numpy(a).to(DEV)
        args = list(map(t, (value, sh, start, loc, attn, go)))
        tiled = _C.ms_deform_attn_backward(*args, 64)
        _C.USE_TILED_BACKWARD = False
        try:
            atomic = _C.ms_deform_attn_backward(*args, 64)
        finally:
            _C.USE_TILED_BACKWARD = True
        want = oracle.msda_backward(go, value, sh, start, loc, attn)
        for a, b, w, name in zip(tiled, atomic, want, ("grad_value", "grad_loc", "grad_attn")):
            _close(a, w, 2e-5, "tiled " + name)
            _close(b, w, 2e-5, "atomic " + name)
        # the two paths fold the D channels in different lane orders: equal up to rounding
        torch.testing.assert_close(tiled[1], atomic[1], rtol=1e-4, atol=1e-5 * float(atomic[1].abs().max()))
        torch.testing.assert_close(tiled[2], atomic[2], rtol=1e-4, atol=1e-5 * float(atomic[2].abs().max()))


def test_tiled_backward_overwrites_poisoned_grad_value():
    """grad_value is written exactly once by the tiled path: no dependence on prior contents
    (the torch.empty buffer the binding hands over) and no row left unwritten."""
    from ziragroundingdino_amd import _lib

    lib = _lib.load()
    B, Q, M, D, shapes, P = 2, 50, 8, 32, NORTH_STAR_SHAPES, 4
    S = sum(h * w for h, w in shapes)
    value, sh, start, loc, attn, go = _random_case(B, Q, M, D, shapes, P, seed=8)
    t = lambda a: torch.from_numpy(a).to(DEV)
    v, tsh, tst, tloc, tattn, tgo = map(t, (value, sh, start, loc, attn, go))
    n = lib.zira_msda_bwd_workspace_bytes(B, S, M, D, 4, Q, P)
    ws = torch.full((n,), 0xAB, dtype=torch.uint8, device=DEV)          # garbage workspace
    gv = torch.full_like(v, float("nan"))
    gl = torch.full_like(tloc, float("nan"))
    ga = torch.full_like(tattn, float("nan"))
    rc = lib.zira_msda_bwd_f32_ws(tgo.data_ptr(), v.data_ptr(), tsh.data_ptr(), tst.data_ptr(),
                                  tloc.data_ptr(), tattn.data_ptr(), B, S, M, D, 4, Q, P,
                                  gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), ws.data_ptr(), n,
                                  torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.isfinite(gv).all() and torch.isfinite(gl).all() and torch.isfinite(ga).all()
    # with Q=50 most value rows receive nothing and must be exactly zero
    assert (gv == 0).float().mean() > 0.5


@pytest.mark.parametrize("fill", [0xAB, 0xFF, 0x00])
def test_dense_backward_with_garbage_workspace(oracle, fill):
    """The dense path (bin + LDS accumulate + fold) through the C ABI: whatever the workspace and the output buffers
    hold before the call -- tickets, per-block maxima, the fallback flag, partial rows, trash rows are all (re)written
    by the call itself -- the result is the oracle's."""
    from ziragroundingdino_amd import _lib

    lib = _lib.load()
    value, sh, start, loc, attn, go = _dense_case(9)
    B, S, M, D = value.shape
    L, P = loc.shape[3], loc.shape[4]
    want = oracle.msda_backward(go, value, sh, start, loc, attn)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    v, tsh, tst, tloc, tattn, tgo = map(t, (value, sh, start, loc, attn, go))
    n = lib.zira_msda_bwd_workspace_bytes(B, S, M, D, L, S, P)
    assert n > 0
    ws = torch.full((n,), fill, dtype=torch.uint8, device=DEV)
    gv = torch.full_like(v, float("nan"))
    gl = torch.full_like(tloc, float("nan"))
    ga = torch.full_like(tattn, float("nan"))
    for _ in range(2):   # (the second call finds what the first one left behind)
        rc = lib.zira_msda_bwd_f32_ws(tgo.data_ptr(), v.data_ptr(), tsh.data_ptr(), tst.data_ptr(),
                                      tloc.data_ptr(), tattn.data_ptr(), B, S, M, D, L, S, P,
                                      gv.data_ptr(), gl.data_ptr(), ga.data_ptr(), ws.data_ptr(), n,
                                      torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        for g, w, name in zip((gv, gl, ga), want, ("grad_value", "grad_loc", "grad_attn")):
            _close(g, w, 5e-5 if name == "grad_value" else 2e-5, name)


# ---- planned backward: the plan is made in the forward pass (zira_msda_plan_f32), the backward takes it as a handle ----
PLANNED_CASES = [("northstar_decoder", dict(lo=0.0, hi=1.0)), ("northstar_inmodel", dict(inmodel=True)),
                 ("northstar_pinpoint", dict(hot=2)), ("oob", dict(lo=-0.5, hi=1.5))]


@pytest.mark.parametrize("name,kw", PLANNED_CASES, ids=[c[0] for c in PLANNED_CASES])
def test_planned_backward_through_the_c_abi(oracle, name, kw):
    """zira_msda_plan_bytes / zira_msda_fwd_plan_f32 / zira_msda_bwd_planned_f32 called directly: a garbage plan buffer,
    the plan written once and used by two backward calls (it is only read), outputs poisoned before each call."""
    from ziragroundingdino_amd import _lib

    lib = _lib.load()
    B, Q, M, D, shapes, P = 2, 900, 8, 32, NORTH_STAR_SHAPES, 4
    S = sum(h * w for h, w in shapes)
    value, sh, start, loc, attn, go = _random_case(B, Q, M, D, shapes, P, seed=31, **kw)
    want_out = oracle.msda_forward(value, sh, start, loc, attn)
    want = oracle.msda_backward(go, value, sh, start, loc, attn)
    t = lambda a: torch.from_numpy(a).to(DEV)
    v, tsh, tst, tloc, tattn, tgo = map(t, (value, sh, start, loc, attn, go))
    n = lib.zira_msda_plan_bytes(B, S, M, D, 4, Q, P)
    assert n > 0 and n == lib.zira_msda_bwd_workspace_bytes(B, S, M, D, 4, Q, P)
    plan = torch.full((n,), 0xAB, dtype=torch.uint8, device=DEV)
    out = torch.full((B, Q, M * D), float("nan"), device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    assert lib.zira_msda_fwd_plan_f32(v.data_ptr(), tsh.data_ptr(), tst.data_ptr(), tloc.data_ptr(), tattn.data_ptr(),
                                      B, S, M, D, 4, Q, P, out.data_ptr(), plan.data_ptr(), n, st) == 0
    _close(out, want_out, 2e-5, "output")
    # the plan alone (zira_msda_plan_f32) serves the same backward
    plan2 = torch.full((n,), 0x5C, dtype=torch.uint8, device=DEV)
    assert lib.zira_msda_plan_f32(tsh.data_ptr(), tst.data_ptr(), tloc.data_ptr(), tattn.data_ptr(), B, S, M, D, 4, Q, P, plan2.data_ptr(), n, st) == 0
    for scale, plan in ((1.0, plan), (-3.0, plan), (1.0, plan2)):
        gv, gl, ga = torch.full_like(v, float("nan")), torch.full_like(tloc, float("nan")), torch.full_like(tattn, float("nan"))
        g = (tgo * scale).contiguous()
        assert lib.zira_msda_bwd_planned_f32(g.data_ptr(), v.data_ptr(), tsh.data_ptr(), tst.data_ptr(), tloc.data_ptr(),
                                             tattn.data_ptr(), B, S, M, D, 4, Q, P, gv.data_ptr(), gl.data_ptr(),
                                             ga.data_ptr(), plan.data_ptr(), n, st) == 0
        torch.cuda.synchronize()
        # (every sample, borders included: the planned kernels form the pixel coordinate like the oracle -- mul, then sub)
        for got, w, nm in zip((gv, gl, ga), want, ("grad_value", "grad_loc", "grad_attn")):
            _close(got, w * scale, 2e-5 * abs(scale), "planned %s (x %g)" % (nm, scale))
    # argument errors are returned, not raised: no planned path for a dense call, a short plan buffer
    assert lib.zira_msda_plan_bytes(B, S, M, D, 4, S, P) == 0
    assert lib.zira_msda_plan_f32(tsh.data_ptr(), tst.data_ptr(), tloc.data_ptr(), tattn.data_ptr(), B, S, M, D, 4, Q, P, plan.data_ptr(), n - 1, st) != 0
    assert lib.zira_msda_bwd_planned_f32(tgo.data_ptr(), v.data_ptr(), tsh.data_ptr(), tst.data_ptr(), tloc.data_ptr(),
                                         tattn.data_ptr(), B, S, M, D, 4, Q, P, gv.data_ptr(), gl.data_ptr(),
                                         ga.data_ptr(), plan.data_ptr(), n - 1, st) != 0


def test_forward_plan_eager_side_stream_and_in_a_graph(oracle):
    """The autograd Function plans right behind the forward gather (`_C.ms_deform_attn_plan`) and hands the plan to the
    backward: same gradients as planning inside the backward call, eagerly, from a side stream, and when forward and
    backward are captured into a hipGraph and replayed on new inputs."""
    B, Q, M, D, shapes, P = 2, 300, 8, 32, [(40, 61), (20, 31), (10, 16), (5, 8)], 4
    value, sh, start, loc, attn, go = _random_case(B, Q, M, D, shapes, P, seed=32, lo=-0.2, hi=1.2)
    t = lambda a: torch.from_numpy(a).to(DEV)
    tsh, tst = t(sh), t(start)
    want = oracle.msda_backward(go, value, sh, start, loc, attn)

    def run(v_, loc_, attn_, go_):
        v_, loc_, attn_ = v_.detach().requires_grad_(), loc_.detach().requires_grad_(), attn_.detach().requires_grad_()
        out = MultiScaleDeformableAttnFunction.apply(v_, tsh, tst, loc_, attn_, 64)
        assert out.grad_fn.plan is not None or not _C.USE_FORWARD_PLAN
        out.backward(go_)
        return out.detach(), v_.grad, loc_.grad, attn_.grad

    got = run(t(value), t(loc), t(attn), t(go))
    _C.USE_FORWARD_PLAN = False
    try:
        ref = run(t(value), t(loc), t(attn), t(go))
    finally:
        _C.USE_FORWARD_PLAN = True
    torch.cuda.synchronize()
    assert torch.equal(got[0], ref[0]) and torch.equal(got[2], ref[2]) and torch.equal(got[3], ref[3])
    torch.testing.assert_close(got[1], ref[1], rtol=1e-5, atol=1e-6 * float(ref[1].abs().max()))
    for g_, w, nm in zip(got[1:], want, ("grad_value", "grad_loc", "grad_attn")):
        _close(g_, w, 2e-5, nm)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        got_s = run(t(value), t(loc), t(attn), t(go))
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(got_s[2], got[2]) and torch.equal(got_s[3], got[3])
    # captured: static inputs, forward + backward in one graph, replayed on other data
    sv, sl, sa, sg = t(value).requires_grad_(), t(loc).requires_grad_(), t(attn).requires_grad_(), t(go)
    warm = torch.cuda.Stream()
    warm.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(warm):
        for _ in range(2):
            MultiScaleDeformableAttnFunction.apply(sv, tsh, tst, sl, sa, 64).backward(sg)
    torch.cuda.current_stream().wait_stream(warm)
    sv.grad = sl.grad = sa.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        MultiScaleDeformableAttnFunction.apply(sv, tsh, tst, sl, sa, 64).backward(sg)
    value2, _, _, loc2, attn2, go2 = _random_case(B, Q, M, D, shapes, P, seed=33, hot=True)
    want2 = oracle.msda_backward(go2, value2, sh, start, loc2, attn2)
    eager2 = run(t(value2), t(loc2), t(attn2), t(go2))
    with torch.no_grad():
        sv.copy_(t(value2)); sl.copy_(t(loc2)); sa.copy_(t(attn2)); sg.copy_(t(go2))
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    errs = []
    for what, got3 in (("eager", eager2[1:]), ("replayed", (sv.grad, sl.grad, sa.grad))):
        for g_, w, nm in zip(got3, want2, ("grad_value", "grad_loc", "grad_attn")):
            g_ = g_.detach().cpu().numpy() if torch.is_tensor(g_) else g_
            errs.append((what, nm, float(np.abs(g_ - w).max()) / max(1.0, float(np.abs(w).max()))))
    assert all(e[2] <= 2e-5 for e in errs), errs


FUSED_SHAPES = [
    # (id, B, Q, M, shapes, P, kwargs): the branches of the fused forward + plan kernel and of the deal
    ("two_pass_p5_heads12", 3, 57, 4, [(12, 9), (6, 5), (3, 3), (2, 2), (1, 1)], 5, {}),
    ("one_group_heads4", 2, 200, 2, [(17, 23)], 1, {}),
    ("two_pass_q1500", 1, 1500, 8, [(30, 41), (15, 21), (8, 11)], 4, dict(clustered=True)),
    ("oob_p3_heads6", 2, 333, 3, [(20, 31), (10, 16), (5, 8), (3, 4)], 3, dict(lo=-0.5, hi=1.5)),
    ("pinpoint_small_map", 2, 900, 8, [(9, 13), (5, 7)], 4, dict(hot=2)),
]


@pytest.mark.parametrize("name,B,Q,M,shapes,P,kw", FUSED_SHAPES, ids=[c[0] for c in FUSED_SHAPES])
def test_fused_forward_plan_and_planned_backward_other_shapes(oracle, name, B, Q, M, shapes, P, kw):
    """`MultiScaleDeformableAttnFunction` on sparse D = 32 calls away from the north-star shape: the fused forward + plan launch
    with the two-pass plan (P > 4 or Q > 1024), fewer than eight heads (one group, every accumulate block deals from all units),
    out-of-window samples (records without corners), everything on a few pixels of a small map (tiles split into many shares)."""
    D = 32
    value, sh, start, loc, attn, go = _random_case(B, Q, M, D, shapes, P, seed=41, **kw)
    t = lambda a: torch.from_numpy(a).to(DEV)
    v, lo, at = t(value).requires_grad_(), t(loc).requires_grad_(), t(attn).requires_grad_()
    assert _C.plan_applies(v, t(sh), t(start), lo, 64)
    out = MultiScaleDeformableAttnFunction.apply(v, t(sh), t(start), lo, at, 64)
    assert out.grad_fn.plan is not None
    out.backward(t(go))
    torch.cuda.synchronize()
    _close(out, oracle.msda_forward(value, sh, start, loc, attn), 2e-5, "output")
    want = oracle.msda_backward(go, value, sh, start, loc, attn)
    _close(v.grad, want[0], 2e-5, "grad_value")
    _close(lo.grad, want[1], 2e-5, "grad_loc")
    _close(at.grad, want[2], 2e-5, "grad_attn")


def test_a_plan_serves_only_the_tensors_it_was_made_from():
    """The plan holds every sample's tile and every record's attention weight: `ms_deform_attn_backward(plan=...)` refuses a
    plan made for other sampling locations / attention weights of the same shape, and one whose tensors were modified in
    place since (the autograd Function's saved tensors are version-checked by autograd; the plan rides beside them)."""
    from ziragroundingdino_amd import _C

    B, Q, M, D, shapes, P = 2, 64, 8, 32, [(12, 17), (6, 9), (3, 5), (2, 3)], 4
    value, sh, start, loc, attn, go = _random_case(B, Q, M, D, shapes, P, seed=41)
    t = lambda a: torch.from_numpy(a).to(DEV)
    v, tsh, tst, tloc, tattn, tgo = map(t, (value, sh, start, loc, attn, go))
    if not _C.plan_applies(v, tsh, tst, tloc, 64):
        pytest.skip("no planned path for this call")
    out, plan = _C.ms_deform_attn_forward_plan(v, tsh, tst, tloc, tattn, 64)
    _C.ms_deform_attn_backward(v, tsh, tst, tloc, tattn, tgo, 64, plan=plan)            # its own tensors: fine
    other = tloc.clone()
    with pytest.raises(RuntimeError, match="plan was made for other"):
        _C.ms_deform_attn_backward(v, tsh, tst, other, tattn, tgo, 64, plan=plan)
    tattn.mul_(0.5)                                                                       # in place: the records hold the old weights
    with pytest.raises(RuntimeError, match="plan was made for other"):
        _C.ms_deform_attn_backward(v, tsh, tst, tloc, tattn, tgo, 64, plan=plan)


@pytest.mark.parametrize("name,kw", [("inmodel", dict(inmodel=True)), ("pinpoint", dict(hot=2)), ("hot", dict(hot=True))])
def test_shares_of_split_tiles_combine_inside_the_launch_under_load(oracle, name, kw):
    """Heavy tiles are cut into shares whose partial tiles meet INSIDE the accumulate launch: write-through stores, an
    agent-scope counter, the last arriver sums them (csrc/msda_tiles.hip; no fold launch since round 5).  A hand-off of this
    kind fails -- if it fails -- rarely, under uneven load, and through STALE copies of lines an earlier call left in a cache:
    so the same plan serves many backward calls that alternate between two gradients (a stale partial tile of the previous
    call would carry the other gradient's sums), while a second stream keeps the memory system busy, and every grad_value
    is compared with the oracle's, element by element."""
    from ziragroundingdino_amd import _C

    B, Q, M, D, shapes, P = 2, 900, 8, 32, NORTH_STAR_SHAPES, 4
    value, sh, start, loc, attn, go = _random_case(B, Q, M, D, shapes, P, seed=53, **kw)
    go2 = np.random.default_rng(54).standard_normal(go.shape).astype(np.float32) * 3.0
    want = [oracle.msda_backward(g, value, sh, start, loc, attn)[0] for g in (go, go2)]
    t = lambda a: torch.from_numpy(a).to(DEV)
    v, tsh, tst, tloc, tattn = map(t, (value, sh, start, loc, attn))
    tgo = [t(go), t(go2)]
    out, plan = _C.ms_deform_attn_forward_plan(v, tsh, tst, tloc, tattn, 64)
    side = torch.cuda.Stream()
    big = torch.empty(64 * 1024 * 1024, device=DEV)          # 256 MB: the copies below evict L2 and the Infinity Cache
    scale = [max(1.0, float(np.abs(w).max())) for w in want]
    for it in range(24):
        if it % 3 != 2:                                       # two calls in three run beside a streaming copy
            with torch.cuda.stream(side):
                big.copy_(big.flip(0) if it % 2 else big + 1.0)
        gv = _C.ms_deform_attn_backward(v, tsh, tst, tloc, tattn, tgo[it & 1], 64, plan=plan)[0]
        err = float((gv - t(want[it & 1])).abs().max()) / scale[it & 1]
        assert err < 2e-5, (name, it, err)
    torch.cuda.synchronize()
