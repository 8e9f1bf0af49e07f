"""The thin products of the fusion block's image side (csrc/thin_f16x2.hip; the re-bracketed BiMultiHeadAttention.forward of
the reference's models/GroundingDINO/fuse_modules.py:170-248 and its autograd).  The accuracy gate: against an fp64 product at
the benchmark's shapes the rms error must not exceed that of the library's fp32 bmm on the same inputs and the maximum error must
stay within 1.25 x of it (at K = 64 both maxima are the rounding of the fp32 result: 2.8e-7 against 2.6e-7 of the scale).  Plus an
exact-integer layout check in both orientations of the small operand, K and N that are not multiples of 32, ragged row counts,
bias and residual, the concatenated contraction with sources 10 orders of magnitude apart, and the autograd wrappers of
dense.py beside the library formulation."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from ziragroundingdino_amd import dense  # noqa: E402


def _ints(shape, g, lo=-8, hi=9):
    return torch.randint(lo, hi, shape, device="cuda", generator=g).float()


@pytest.mark.parametrize("B,M,K,N", [(2, 130, 256, 64), (2, 130, 64, 256), (1, 77, 52, 100), (3, 4097, 128, 36), (2, 64, 96, 2048),
                                      (2, 333, 4, 4), (2, 40, 32, 32)])
def test_exact_on_small_integers_both_orientations(B, M, K, N):
    g = torch.Generator(device="cuda").manual_seed(1)
    a, w = _ints((B, M, K), g), _ints((B, K, N), g) + (torch.arange(N, device="cuda") % 3)
    want = torch.bmm(a.double(), w.double()).float()
    assert torch.equal(dense.thin_bmm(a, w, True), want)
    assert torch.equal(dense.thin_bmm(a, w.transpose(1, 2).contiguous(), False), want)
    bias, res = _ints((B, N), g), _ints((B, M, N), g)
    assert torch.equal(dense.thin_bmm(a, w, True, bias=bias, res=res), want + bias[:, None] + res)
    if K <= 128:
        a2, w2 = _ints((B, M, K), g), _ints((B, K, N), g)
        want2 = want + torch.bmm(a2.double(), w2.double()).float()
        assert torch.equal(dense.thin_bmm(a, w, True, A2=a2, W2=w2), want2)
        acc = res.clone()
        out = dense.thin_bmm(a, w, True, A2=a2, W2=w2, res=acc, out=acc)          # the residual may be the output
        assert out.data_ptr() == acc.data_ptr() and torch.equal(acc, want2 + res)


@pytest.mark.parametrize("K,N,what", [(256, 64, "scores, 16 text tokens per image"), (256, 128, "scores, 32 text tokens"),
                                       (64, 256, "image output, 16 text tokens"), (128, 256, "image output, 32 text tokens")])
def test_accuracy_gate_against_fp64_beside_the_library_bmm(K, N, what):
    torch.manual_seed(2)
    B, M = 2, 22223
    a = torch.randn(B, M, K, device="cuda")
    if K < 256:
        a = a.softmax(-1)                                  # (attention probabilities on the K side)
    w = torch.randn(B, K, N, device="cuda") * 0.05
    ref = torch.bmm(a.double(), w.double())
    lib = torch.bmm(a, w).double()
    ours = dense.thin_bmm(a, w, True).double()
    scale = float(ref.abs().max())
    e_lib, e_ours = (lib - ref).abs(), (ours - ref).abs()
    stats = "max %.3e / %.3e, rms %.3e / %.3e of the scale (ours / library)" % (
        float(e_ours.max()) / scale, float(e_lib.max()) / scale, float(e_ours.pow(2).mean().sqrt()) / scale,
        float(e_lib.pow(2).mean().sqrt()) / scale)
    print(what, stats)
    assert float(e_ours.max()) <= 1.25 * float(e_lib.max()), stats
    assert float(e_ours.pow(2).mean().sqrt()) <= float(e_lib.pow(2).mean().sqrt()), stats


def test_concatenated_sources_keep_their_own_scales():
    """g_v = e g_u + g_xm a^T: probabilities (<= 1) against gradients of 1e-7 in one pass; each source's error is measured
    against ITS OWN product's size."""
    torch.manual_seed(3)
    B, M, K, N = 2, 22223, 64, 256
    e, gu = torch.rand(B, M, K, device="cuda"), torch.randn(B, K, N, device="cuda") * 1e-9
    gx, at = torch.randn(B, M, K, device="cuda") * 1e-8, torch.randn(B, K, N, device="cuda") * 3.0
    p1, p2 = torch.bmm(e.double(), gu.double()), torch.bmm(gx.double(), at.double())
    ours = dense.thin_bmm(e, gu, True, A2=gx, W2=at).double()
    lib = (torch.bmm(e, gu) + torch.bmm(gx, at)).double()
    bound = lambda x: float((x - (p1 + p2)).abs().max())
    assert bound(ours) <= max(bound(lib), 2e-7 * float((p1.abs() + p2.abs()).max())), (bound(ours), bound(lib))
    # and with the first source switched off by a zero matrix the second one alone is as good as a single-source call
    alone = dense.thin_bmm(gx, at, True).double()
    both = dense.thin_bmm(e, torch.zeros_like(gu), True, A2=gx, W2=at).double()
    assert torch.equal(alone, both)


def test_magnitudes_over_24_orders():
    torch.manual_seed(4)
    B, M, K, N = 2, 4096, 256, 64
    a = torch.randn(B, M, K, device="cuda") * torch.logspace(-12, 12, M, device="cuda")[None, :, None]
    w = torch.randn(B, N, K, device="cuda") * torch.logspace(-6, 6, N, device="cuda")[None, :, None]
    ref = torch.bmm(a.double(), w.double().transpose(1, 2))
    absref = torch.bmm(a.double().abs(), w.double().abs().transpose(1, 2))
    ours = dense.thin_bmm(a, w, False).double()
    lib = torch.bmm(a, w.transpose(1, 2)).double()
    assert float(((ours - ref).abs() / absref).max()) <= max(float(((lib - ref).abs() / absref).max()), 3e-7)


def test_rows_past_the_end_and_poisoned_output():
    torch.manual_seed(5)
    for M in (2049, 2079, 2080):
        a, w = torch.randn(2, M, 64, device="cuda"), torch.randn(2, 64, 256, device="cuda")
        buf = torch.full((2 * M * 256 + 1024,), float("nan"), device="cuda")
        out = buf[: 2 * M * 256].view(2, M, 256)
        dense.thin_bmm(a, w, True, out=out)
        assert bool(torch.isfinite(out).all()) and bool(torch.isnan(buf[2 * M * 256:]).all())
        assert float((out.double() - torch.bmm(a.double(), w.double())).abs().max()) < 2e-5


def test_unsupported_arguments_are_refused():
    from ziragroundingdino_amd import _lib
    lib = _lib.load()
    assert lib.zira_thin_f16x2_frag_bytes(64, 192) == 0 and lib.zira_thin_f16x2_frag_bytes(64, 256) > 0
    a = torch.zeros(1, 64, 192, device="cuda")
    assert not dense.thin_supported(torch.zeros(1, 4096, 192, device="cuda"), 64, 192)
    assert dense.thin_supported(torch.zeros(1, 4096, 256, device="cuda"), 64, 256)
    assert not dense.thin_supported(torch.zeros(1, 4096, 62, device="cuda"), 64, 62)
    f = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    assert lib.zira_thin_f16x2_f32(a.data_ptr(), f.data_ptr(), None, None, 1, 64, 64, 192, None, None, a.data_ptr(), None) == -1
    assert lib.zira_thin_f16x2_f32(a.data_ptr(), f.data_ptr(), a.data_ptr(), None, 1, 64, 64, 64, None, None, a.data_ptr(), None) == -1
    assert lib.zira_thin_f16x2_f32(a.data_ptr(), f.data_ptr(), a.data_ptr(), f.data_ptr(), 1, 64, 64, 256, None, None, a.data_ptr(), None) == -1


def test_autograd_wrappers_beside_the_library_formulation():
    """wide_matmul, wide_matmul_residual and tall_reduce_nt at the fusion block's shapes: forward and every gradient against an
    fp64 evaluation of the same expressions, beside the same wrappers with the thin kernels switched off."""
    torch.manual_seed(6)
    B, M, C, n = 2, 22223, 256, 128      # (32 text tokens per image: the row GEMM of the switched-off path needs K >= 128)
    v = torch.randn(B, M, C, device="cuda")
    a = torch.randn(B, C, n, device="cuda") * 0.05
    z = torch.randn(B, n, C, device="cuda") * 0.05
    bias, scale = torch.randn(C, device="cuda"), torch.rand(C, device="cuda") * 1e-2
    gout, gu = torch.randn(B, M, C, device="cuda"), torch.randn(B, n, C, device="cuda")

    def run(dtype, thin):
        dense.USE_THIN = thin
        try:
            vv, aa, zz = (t.detach().to(dtype).clone().requires_grad_(True) for t in (v, a, z))
            if dtype == torch.float64:
                xm = torch.bmm(vv, aa)
                e = xm.softmax(-1)
                out = vv + scale.double() * (torch.bmm(e, zz) + bias.double())
                u = torch.bmm(e.transpose(1, 2), vv)
            else:
                xm = dense.wide_matmul(vv, aa)
                e = xm.softmax(-1)
                out = dense.wide_matmul_residual(e, zz, bias, vv, scale)
                u = dense.tall_reduce_nt(e, vv)
            ((out * gout.to(dtype)).sum() + (u * gu.to(dtype)).sum()).backward()
            return [t.detach().double() for t in (xm, out, u, vv.grad, aa.grad, zz.grad)]
        finally:
            dense.USE_THIN = True

    before = dense.thin_bmm.calls
    ours = run(torch.float32, True)
    assert dense.thin_bmm.calls - before == 6          # scores, image output, and the four [M, *] gradients
    lib, ref = run(torch.float32, False), run(torch.float64, False)
    for name, o, l, r in zip(("xm", "out", "u", "g_v", "g_a", "g_z"), ours, lib, ref):
        s = float(r.abs().max())
        eo, el = float((o - r).abs().max()) / s, float((l - r).abs().max()) / s
        print("%-4s max error / scale: thin %.3e, library %.3e" % (name, eo, el))
        assert eo <= max(1.5 * el, 2e-6), name


@pytest.mark.parametrize("N,T,masked", [(22223, 16, False), (22223, 32, True), (2500, 7, True)])
def test_fused_image_side_node_beside_the_four_ops_it_holds(N, T, masked):
    """dense._FusionImageSide against wide_matmul -> bi_softmax -> tall_reduce_nt / wide_matmul_residual on the same inputs: the
    forward launches the same kernels (bit-equal); the gradients differ only in the order in which the three contributions to
    g_vn are added (one pass instead of two accumulations)."""
    torch.manual_seed(7)
    B, C, H = 2, 256, 4
    n = H * T
    vn = torch.randn(B, N, C, device="cuda")
    a, c = torch.randn(B, C, n, device="cuda") * 0.05, torch.randn(B, n, device="cuda") * 0.1
    z = torch.randn(B, n, C, device="cuda") * 0.05
    bias, scale = torch.randn(C, device="cuda"), torch.rand(B, 1, C, device="cuda") * 1e-2
    ml = mv = None
    if masked:
        ml = torch.zeros(B, T, dtype=torch.bool, device="cuda"); ml[1, T - 2:] = True
        mv = torch.zeros(B, N, dtype=torch.bool, device="cuda"); mv[0, N - 100:] = True
    g_out, g_t, g_cs = torch.randn(B, N, C, device="cuda"), torch.randn(B, n, C, device="cuda") * 1e-3, torch.randn(B, n, device="cuda") * 1e-3
    assert dense.fusion_image_side_supported(vn, a, z, bias, scale, H, T, False)

    def run(fused):
        leaves = [t.detach().clone().requires_grad_(True) for t in (vn, a, c, z)]
        v_, a_, c_, z_ = leaves
        if fused:
            out, t, colsum = dense.fusion_image_side(v_, a_, c_, z_, bias, scale, ml, mv, H, T)
        else:
            xm = dense.wide_matmul(v_, a_)
            pv, e, colsum = dense.bi_softmax(xm, c_, ml, mv, H, T)
            t = dense.tall_reduce_nt(e, v_)
            out = dense.wide_matmul_residual(pv, z_, bias, v_, scale)
        # (e and colsum must be used as e / colsum: the text output divides)
        ((out * g_out).sum() + ((t / colsum[..., None]) * g_t).sum()).backward()
        return [out.detach(), t.detach(), colsum.detach()] + [x.grad for x in leaves]

    before = dense.thin_bmm.calls
    got = run(True)
    assert dense.thin_bmm.calls - before == 5           # scores, output; g_pv, g_e, and g_vn in one pass
    want = run(False)
    for name, g, w in zip(("out", "t", "colsum", "g_vn", "g_a", "g_c", "g_z"), got, want):
        if name in ("out", "t", "colsum"):
            assert torch.equal(g, w), name
        else:
            s = float(w.abs().max())
            assert float((g - w).abs().max()) <= 2e-6 * s, (name, float((g - w).abs().max()) / s)
