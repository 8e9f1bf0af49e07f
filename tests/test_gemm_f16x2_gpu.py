"""fp32-accurate GEMM in two-plane f16 arithmetic (csrc/gemm_f16x2.hip; reference products: the 256-wide projections of
ms_deform_attn.py:262-288 and the backbone's linears under the freeze of groundingdino_dual_zero_rep_branch.py:722-745).
The accuracy gate: against an fp64 product on the model's own shapes, the maximum and the rms error must not exceed those of
the library's fp32 GEMM on the same inputs.  Plus an exact-integer layout check, the four epilogues, ragged row counts,
magnitudes over 24 orders and the cached helpers under ``Switches.gemm_arith = "f16x2"``."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from ziragroundingdino_amd import gemm_bf16x3 as g3  # noqa: E402


def _ref64(a, w_nk):
    return a.double() @ w_nk.double().t()


def test_exact_on_small_integers_with_asymmetric_operands():
    g = torch.Generator(device="cuda").manual_seed(1)
    for M, N, K in ((130, 128, 32), (257, 256, 96), (200, 384, 64)):
        a = torch.randint(-8, 9, (M, K), device="cuda", generator=g).float()
        w = torch.randint(-8, 9, (N, K), device="cuda", generator=g).float() + torch.arange(N, device="cuda")[:, None] % 3
        bias = torch.randint(-4, 5, (N,), device="cuda", generator=g).float()
        want = a @ w.t() + bias
        got = g3.gemm_f16x2(a, g3.split_planes_f16x2(w, False), N, g3.EPI_BIAS, bias=bias)
        assert torch.equal(got, want), (M, N, K)
        got_t = g3.gemm_f16x2(a, g3.split_planes_f16x2(w.t().contiguous(), True), N, g3.EPI_BIAS, bias=bias)   # the weight stored [K, N]
        assert torch.equal(got_t, want), (M, N, K)


@pytest.mark.parametrize("M,N,K,what", [(44446, 256, 256, "the 256-wide projections"), (44446, 384, 256, "the query projection"),
                                         (44446, 256, 384, "its input gradient"), (33600, 768, 192, "a Swin stage-2 fc1"),
                                         (44446, 256, 2048, "K = 2048 on post-ReLU operands")])
def test_accuracy_gate_against_fp64_beside_the_library_fp32_gemm(M, N, K, what):
    torch.manual_seed(2)
    a = torch.randn(M, K, device="cuda")
    if K == 2048:
        a = a.relu_()
    w = torch.randn(N, K, device="cuda") * 0.05
    ref = _ref64(a, w)
    lib = (a @ w.t()).double()
    ours = g3.gemm_f16x2(a, g3.split_planes_f16x2(w, False), N, g3.EPI_ADD, aux=torch.zeros(M, N, device="cuda")).double()
    scale = float(ref.abs().max())
    e_lib, e_ours = (lib - ref).abs(), (ours - ref).abs()
    stats = "max %.3e / %.3e, rms %.3e / %.3e of the scale (ours / library)" % (
        float(e_ours.max()) / scale, float(e_lib.max()) / scale, float(e_ours.pow(2).mean().sqrt()) / scale,
        float(e_lib.pow(2).mean().sqrt()) / scale)
    print(what, stats)
    assert float(e_ours.max()) <= float(e_lib.max()), stats
    assert float(e_ours.pow(2).mean().sqrt()) <= float(e_lib.pow(2).mean().sqrt()), stats


def test_accuracy_over_magnitudes_and_cancellation():
    torch.manual_seed(3)
    M, N, K = 1024, 128, 256
    a = torch.randn(M, K, device="cuda") * torch.logspace(-12, 12, M, device="cuda")[:, None]
    a[:, ::7] *= 1e-3                                   # columns of very different size inside a row's K step
    w = torch.randn(N, K, device="cuda") * torch.logspace(-6, 6, N, device="cuda")[:, None]
    ref, absref = _ref64(a, w), a.double().abs() @ w.double().abs().t()
    ours = g3.gemm_f16x2(a, g3.split_planes_f16x2(w, False), N, g3.EPI_BIAS, bias=torch.zeros(N, device="cuda")).double()
    lib = (a @ w.t()).double()
    assert float(((ours - ref).abs() / absref).max()) <= max(float(((lib - ref).abs() / absref).max()), 3e-7)


@pytest.mark.parametrize("M", [1, 127, 192, 1000])
def test_epilogues_and_ragged_rows(M):
    torch.manual_seed(4)
    N, K = 256, 64
    a, w = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    bias, aux = torch.randn(N, device="cuda"), torch.randn(M, N, device="cuda")
    planes = g3.split_planes_f16x2(w, False)
    prod = (a.double() @ w.double().t())
    close = lambda x, y: float((x.double() - y).abs().max()) <= 2e-6 * max(1.0, float(y.abs().max()))
    assert close(g3.gemm_f16x2(a, planes, N, g3.EPI_BIAS, bias=bias), prod + bias.double())
    assert close(g3.gemm_f16x2(a, planes, N, g3.EPI_BIAS_RELU, bias=bias), (prod + bias.double()).relu())
    masked = g3.gemm_f16x2(a, planes, N, g3.EPI_MASK, aux=aux)
    assert close(masked, torch.where(aux > 0, prod, torch.zeros_like(prod))) and bool((masked[aux <= 0] == 0).all())
    acc = aux.clone()
    out = g3.gemm_f16x2(a, planes, N, g3.EPI_ADD, aux=acc, out=acc)
    assert out.data_ptr() == acc.data_ptr() and close(acc, prod + aux.double())
    guard = torch.full((M + 8, N), 7.0, device="cuda")
    g3.gemm_f16x2(a, planes, N, g3.EPI_BIAS, bias=bias, out=guard[:M])
    assert bool((guard[M:] == 7.0).all())


def test_cached_helpers_take_the_two_plane_form_under_f16x2_and_follow_the_weight(monkeypatch):
    from ziragroundingdino_amd import transformer as zt
    monkeypatch.setattr(zt.Switches, "gemm_arith", "f16x2")
    calls = []
    real = g3.gemm_f16x2
    monkeypatch.setattr(g3, "gemm_f16x2", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    owner = torch.nn.Linear(128, 128).cuda()
    owner.weight.requires_grad_(False)
    x = torch.randn(300, 128, device="cuda")
    y = g3.linear(owner, "w", x, owner.weight, owner.bias.detach())
    assert len(calls) == 1 and float((y.double() - (x.double() @ owner.weight.double().t() + owner.bias.detach().double())).abs().max()) <= 1e-5
    gx = g3.linear_input_grad(owner, "w", y, owner.weight)
    assert len(calls) == 2 and float((gx.double() - y.double() @ owner.weight.double()).abs().max()) <= 1e-4
    acc = torch.ones_like(x)
    g3.linear_input_grad(owner, "w", y, owner.weight, accumulate_into=acc)
    assert float((acc.double() - 1 - y.double() @ owner.weight.double()).abs().max()) <= 1e-4
    ptrs = {k: sw.buf.data_ptr() for k, sw in owner.__dict__["_bf16x3_split"].items()}
    with torch.no_grad():
        owner.weight.mul_(2.0)
    g3.refresh(owner, {"w": owner.weight})
    assert {k: sw.buf.data_ptr() for k, sw in owner.__dict__["_bf16x3_split"].items()} == ptrs      # in place
    y2 = g3.linear(owner, "w", x, owner.weight, owner.bias.detach())
    assert float((y2.double() - (x.double() @ owner.weight.double().t() + owner.bias.detach().double())).abs().max()) <= 1e-5


# ---- the panel kernel (csrc/gemm_f16x2_panel.hip): K = 256 / 384, a block = 32 rows and all of K ------------------------------------

def test_panel_exact_on_small_integers_and_the_added_operand():
    g = torch.Generator(device="cuda").manual_seed(5)
    for M, N, K in ((1, 32, 256), (130, 96, 256), (257, 384, 256), (200, 256, 384)):
        a = torch.randint(-8, 9, (M, K), device="cuda", generator=g).float()
        a2 = torch.randint(-8, 9, (M, K), device="cuda", generator=g).float()
        w = torch.randint(-8, 9, (N, K), device="cuda", generator=g).float() + torch.arange(N, device="cuda")[:, None] % 3
        bias = torch.randint(-4, 5, (N,), device="cuda", generator=g).float()
        fr = g3.split_frags_f16x2(w, False)
        assert torch.equal(g3.gemm_f16x2_panel(a, fr, N, g3.EPI_BIAS, bias=bias), a @ w.t() + bias), (M, N, K)
        assert torch.equal(g3.gemm_f16x2_panel(a, fr, N, g3.EPI_BIAS, bias=bias, add=a2), (a + a2) @ w.t() + bias), (M, N, K)
        fr_t = g3.split_frags_f16x2(w.t().contiguous(), True)                       # the weight stored [K, N]
        assert torch.equal(fr_t, fr)
        guard = torch.full((M + 8, N), 7.0, device="cuda")
        g3.gemm_f16x2_panel(a, fr, N, g3.EPI_BIAS, bias=bias, out=guard[:M])
        assert bool((guard[M:] == 7.0).all())


@pytest.mark.parametrize("M,N,K,what", [(44446, 256, 256, "value / output projection"), (44446, 384, 256, "query projection"),
                                         (44446, 256, 384, "its input gradient")])
def test_panel_accuracy_gate_against_fp64_beside_the_library_fp32_gemm(M, N, K, what):
    torch.manual_seed(12)
    a = torch.randn(M, K, device="cuda") * torch.logspace(-2, 2, M, device="cuda")[:, None]
    w = torch.randn(N, K, device="cuda") * 0.05
    ref, absref = _ref64(a, w), a.double().abs() @ w.double().abs().t()
    lib = (a @ w.t()).double()
    ours = g3.gemm_f16x2_panel(a, g3.split_frags_f16x2(w, False), N, g3.EPI_ADD, aux=torch.zeros(M, N, device="cuda")).double()
    e_lib, e_ours = ((lib - ref).abs() / absref), ((ours - ref).abs() / absref)     # (rows of very different size: relative to sum |a b|)
    stats = "max %.3e / %.3e, rms %.3e / %.3e of sum |a b| (ours / library)" % (
        float(e_ours.max()), float(e_lib.max()), float(e_ours.pow(2).mean().sqrt()), float(e_lib.pow(2).mean().sqrt()))
    print(what, stats)
    assert float(e_ours.max()) <= float(e_lib.max()), stats
    assert float(e_ours.pow(2).mean().sqrt()) <= float(e_lib.pow(2).mean().sqrt()), stats


def test_panel_epilogues_and_the_cached_helpers_choose_it(monkeypatch):
    from ziragroundingdino_amd import transformer as zt
    torch.manual_seed(4)
    M, N, K = 1000, 256, 256
    a, w = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    bias, aux = torch.randn(N, device="cuda"), torch.randn(M, N, device="cuda")
    fr = g3.split_frags_f16x2(w, False)
    prod = a.double() @ w.double().t()
    close = lambda x, y: float((x.double() - y).abs().max()) <= 2e-6 * max(1.0, float(y.abs().max()))
    assert close(g3.gemm_f16x2_panel(a, fr, N, g3.EPI_BIAS_RELU, bias=bias), (prod + bias.double()).relu())
    masked = g3.gemm_f16x2_panel(a, fr, N, g3.EPI_MASK, aux=aux)
    assert close(masked, torch.where(aux > 0, prod, torch.zeros_like(prod))) and bool((masked[aux <= 0] == 0).all())
    acc = aux.clone()
    g3.gemm_f16x2_panel(a, fr, N, g3.EPI_ADD, aux=acc, out=acc)
    assert close(acc, prod + aux.double())
    monkeypatch.setattr(zt.Switches, "gemm_arith", "f16x2")
    calls = []
    real = g3.gemm_f16x2_panel
    monkeypatch.setattr(g3, "gemm_f16x2_panel", lambda *a_, **k: (calls.append(1), real(*a_, **k))[1])
    owner = torch.nn.Linear(256, 384).cuda()
    owner.weight.requires_grad_(False)
    x, pos = torch.randn(300, 256, device="cuda"), torch.randn(300, 256, device="cuda")
    y = g3.linear(owner, "w", x, owner.weight, owner.bias.detach(), add=pos)
    assert len(calls) == 1 and close(y, (x + pos).double() @ owner.weight.double().t() + owner.bias.detach().double())
    gx = g3.linear_input_grad(owner, "w", y, owner.weight)                  # K = 384 -> N = 256
    assert len(calls) == 2 and float((gx.double() - y.double() @ owner.weight.double()).abs().max()) <= 1e-3


@pytest.mark.parametrize("M,N,K", [(300, 96, 96), (1000, 288, 96), (259, 192, 768), (130, 576, 192), (8400, 1152, 384)])
def test_widths_that_are_multiples_of_32_only(M, N, K):
    """Swin-T's 96 / 192 / 288 / 576-wide linears: the last column tile is partly outside the matrix."""
    g = torch.Generator(device="cuda").manual_seed(11)
    a = torch.randint(-8, 9, (M, K), device="cuda", generator=g).float()
    w = torch.randint(-8, 9, (N, K), device="cuda", generator=g).float() + torch.arange(N, device="cuda")[:, None] % 3
    bias = torch.randint(-4, 5, (N,), device="cuda", generator=g).float()
    buf = torch.full((M * N + 4096,), float("nan"), device="cuda")
    out = buf[: M * N].view(M, N)
    got = g3.gemm_f16x2(a, g3.split_planes_f16x2(w, False), N, g3.EPI_BIAS, bias=bias, out=out)
    assert torch.equal(got, a @ w.t() + bias) and bool(torch.isnan(buf[M * N:]).all())


@pytest.mark.parametrize("M,N,K,rows", [(4000, 384, 96, 2000), (1000, 96, 384, 250), (777, 192, 192, 777), (9000, 1536, 384, 4500)])
def test_gelu_and_scaled_residual_epilogues(M, N, K, rows):
    """The backbone's block epilogues (reference swin_transformer.py:40-62, :237-262): exact GELU of (product + bias) -- the same
    bits as ATen's GELU of this kernel's own bias epilogue -- and residual + scale[image] * (product + bias) beside addcmul."""
    torch.manual_seed(12)
    a, w, bias = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * 0.1, torch.randn(N, device="cuda")
    planes = g3.split_planes_f16x2(w, False)
    lin = g3.gemm_f16x2(a, planes, N, g3.EPI_BIAS, bias=bias)
    assert float((lin.double() - (a.double() @ w.double().t() + bias.double())).abs().max()) < 2e-5
    assert torch.equal(g3.gemm_f16x2(a, planes, N, g3.EPI_BIAS_GELU, bias=bias), torch.nn.functional.gelu(lin))
    res = torch.randn(M, N, device="cuda")
    scale = torch.tensor([0.0, 1.25, 1.0, 2.5], device="cuda")[: (M + rows - 1) // rows].contiguous()
    got = g3.gemm_f16x2(a, planes, N, g3.EPI_BIAS_RES, bias=bias, aux=res, row_scale=scale, rows_per_scale=rows)
    want = torch.addcmul(res, lin, scale.repeat_interleave(rows)[:M, None])
    assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max())
    assert torch.equal(got[:rows], res[:rows])                                  # a dropped branch leaves the residual untouched
    plain = g3.gemm_f16x2(a, planes, N, g3.EPI_BIAS_RES, bias=bias, aux=res)     # no scale: residual + product
    assert torch.equal(plain, res + lin)
    acc = res.clone()
    g3.gemm_f16x2(a, planes, N, g3.EPI_BIAS_RES, bias=bias, aux=acc, out=acc)    # in place over the residual
    assert torch.equal(acc, plain)


def test_swin_block_with_fused_epilogues_beside_the_composition():
    """One frozen Swin-T stage-1 block pair at 8k+ tokens: GELU and both residuals inside the GEMMs against GEMM + ATen ops
    (same products: only the residual's multiply-add may round differently), in training mode (stochastic depth drawn by the
    caller) and in eval mode."""
    from ziragroundingdino_amd import backbone as bb
    from ziragroundingdino_amd import transformer as zt
    assert zt.Switches.gemm_arith == "f16x2"
    torch.manual_seed(13)
    blocks = torch.nn.ModuleList([bb.SwinTransformerBlock(96, 3, 7, 0, 4.0, 0.1), bb.SwinTransformerBlock(96, 3, 7, 3, 4.0, 0.1)]).cuda()
    for p in blocks.parameters():
        p.requires_grad_(False)
    B, H, W = 2, 70, 63
    x = torch.randn(B, H * W, 96, device="cuda")
    dp = torch.tensor([[[1.0, 0.0], [1.25, 1.25]], [[1.25, 1.25], [0.0, 1.25]]], device="cuda").view(2, 2, B, 1, 1)

    def run(fused, train):
        bb.FUSED_EPILOGUES = fused
        blocks.train(train)
        try:
            with torch.no_grad():
                y = x
                for i, blk in enumerate(blocks):
                    y = blk(y, H, W, None, dp[i] if train else None)
            return y
        finally:
            bb.FUSED_EPILOGUES = True

    for train in (True, False):
        before = g3.CALLS["gemm_f16x2"]
        got = run(True, train)
        assert g3.CALLS["gemm_f16x2"] - before == 8           # qkv, proj, fc1, fc2 of both blocks: the widths 288 and 96 included
        want = run(False, train)
        assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max()), train


@pytest.mark.parametrize("M,N,K", [(300, 256, 776), (1000, 128, 36), (22223, 256, 776), (515, 96, 100)])
def test_contraction_lengths_that_are_multiples_of_4_only(M, N, K):
    """K = 776 (4 heads x 194 text tokens on the contraction side of the fusion block): the planes' rows are padded with zeros to
    whole 32-deep steps, the activation's missing columns are zeros.  Exact on small integers; beside fp64 on random data."""
    g = torch.Generator(device="cuda").manual_seed(21)
    a = torch.randint(-8, 9, (M, K), device="cuda", generator=g).float()
    w = torch.randint(-8, 9, (N, K), device="cuda", generator=g).float()
    bias = torch.randint(-4, 5, (N,), device="cuda", generator=g).float()
    want = (a.double() @ w.double().t()).float() + bias
    assert torch.equal(g3.gemm_f16x2(a, g3.split_planes_f16x2(w, False), N, g3.EPI_BIAS, bias=bias), want)
    assert torch.equal(g3.gemm_f16x2(a, g3.split_planes_f16x2(w.t().contiguous(), True), N, g3.EPI_BIAS, bias=bias), want)
    a, w = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda") * 0.05
    ref = a.double() @ w.double().t()
    ours = g3.gemm_f16x2(a, g3.split_planes_f16x2(w, False), N, g3.EPI_BIAS, bias=torch.zeros(N, device="cuda")).double()
    lib = (a @ w.t()).double()
    assert float((ours - ref).abs().max()) <= 1.25 * float((lib - ref).abs().max())


def test_long_text_side_on_the_contraction_index_through_the_autograd_wrappers():
    """dense.wide_matmul / wide_matmul_residual / tall_reduce_nt at 194 text tokens (H T = 776): the K = 776 products leave the
    library for one tiled two-plane GEMM per image; values and gradients beside the library formulation."""
    from ziragroundingdino_amd import dense
    torch.manual_seed(22)
    B, M, C, n = 2, 5003, 256, 776
    v = torch.randn(B, M, C, device="cuda")
    a, z = torch.randn(B, C, n, device="cuda") * 0.05, torch.randn(B, n, C, device="cuda") * 0.05
    bias, scale = torch.randn(C, device="cuda"), torch.rand(C, device="cuda") * 1e-2
    gout, gu = torch.randn(B, M, C, device="cuda"), torch.randn(B, n, C, device="cuda")

    def run(on):
        dense.USE_THIN = on
        try:
            vv, aa, zz = (t.detach().clone().requires_grad_(True) for t in (v, a, z))
            xm = dense.wide_matmul(vv, aa)
            e = xm.softmax(-1)
            if dense.wide_matmul_residual_supported(e, zz, bias, vv, scale):
                out = dense.wide_matmul_residual(e, zz, bias, vv, scale)
            else:                                    # (what the module does then: the residual as a pass of its own)
                out = torch.addcmul(vv, dense.wide_matmul(e, zz, bias), scale)
            u = dense.tall_reduce_nt(e, vv)
            ((out * gout).sum() + (u * gu).sum()).backward()
            return [t.detach().double() for t in (xm, out, u, vv.grad, aa.grad, zz.grad)]
        finally:
            dense.USE_THIN = True

    before = dense.wide_k_bmm.calls
    ours = run(True)
    assert dense.wide_k_bmm.calls - before == 3          # the image output, and the two [M, 776] x [776, 256] gradients
    lib = run(False)
    for name, o, l in zip(("xm", "out", "u", "g_v", "g_a", "g_z"), ours, lib):
        assert float((o - l).abs().max()) <= 3e-6 * float(l.abs().max()), name
