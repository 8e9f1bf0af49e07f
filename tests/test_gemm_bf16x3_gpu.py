"""fp32-accurate GEMM on the bf16 matrix cores (csrc/gemm_bf16x3.hip; reference products: the FFN of
transformer_for_adapter.py:877-886 and its backward under the freeze of groundingdino_dual_zero_rep_branch.py:722-745).
The accuracy gate: against an fp64 product on the model's own shapes, the maximum and the rms error must not exceed those of
the library's fp32 GEMM (what ``F.linear`` runs) on the same inputs.  Plus an exact-integer layout check (asymmetric operands),
the four epilogues, ragged row counts and the in-place weight refresh."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from ziragroundingdino_amd import gemm_bf16x3 as g3  # noqa: E402


def _ref64(a, w_nk):
    return a.double() @ w_nk.double().t()


def test_split_is_exact_and_planes_are_bfloat16():
    torch.manual_seed(0)
    w = torch.randn(96, 160, device="cuda") * torch.logspace(-6, 6, 160, device="cuda")
    for transpose in (False, True):
        p = g3.split_planes(w, transpose)
        f = (p.to(torch.int32) << 16).view(torch.float32)          # bf16 bits -> fp32
        src = w.t() if transpose else w
        assert torch.equal(f[0] + f[1] + f[2], src)                 # a1 + a2 + a3 == a, exactly (each partial sum is exact too)
        assert float((f[1].abs() / src.abs().clamp_min(1e-30)).max()) <= 2.0 ** -8
        assert float((f[2].abs() / src.abs().clamp_min(1e-30)).max()) <= 2.0 ** -16


def test_exact_on_small_integers_with_asymmetric_operands():
    """Integers up to 2^7 are bfloat16 numbers and their products / sums are exact in fp32: any fragment-layout or
    transposition mistake shows as a wrong integer (A = I with a symmetric B would hide one)."""
    g = torch.Generator(device="cuda").manual_seed(1)
    for M, N, K in ((130, 128, 32), (257, 256, 96), (200, 384, 64)):
        a = torch.randint(-8, 9, (M, K), device="cuda", generator=g).float()
        w = torch.randint(-8, 9, (N, K), device="cuda", generator=g).float() + torch.arange(N, device="cuda")[:, None] % 3
        bias = torch.randint(-4, 5, (N,), device="cuda", generator=g).float()
        want = a @ w.t() + bias
        got = g3.gemm(a, g3.split_planes(w, False), g3.EPI_BIAS, bias=bias)
        assert torch.equal(got, want), (M, N, K)
        got_t = g3.gemm(a, g3.split_planes(w.t().contiguous(), True), g3.EPI_BIAS, bias=bias)   # the weight stored [K, N]
        assert torch.equal(got_t, want), (M, N, K)


@pytest.mark.parametrize("M,N,K,what", [(44446, 2048, 256, "FFN linear1 / the dReLU product: 128-row tiles"),
                                         (44446, 256, 2048, "FFN linear2 / the input gradient: 192-row tiles"),
                                         (44446, 256, 256, "the 256-wide projections")])
def test_accuracy_gate_against_fp64_beside_the_library_fp32_gemm(M, N, K, what):
    torch.manual_seed(2)
    a = torch.randn(M, K, device="cuda")
    if K == 2048:
        a = a.relu_()                       # the second FFN product reads post-ReLU activations
    w = torch.randn(N, K, device="cuda") * 0.05
    ref = _ref64(a, w)
    lib = (a @ w.t()).double()
    ours = g3.gemm(a, g3.split_planes(w, False), g3.EPI_ADD, aux=torch.zeros(M, N, device="cuda")).double()
    scale = float(ref.abs().max())
    e_lib, e_ours = (lib - ref).abs(), (ours - ref).abs()
    stats = "max %.3e / %.3e, rms %.3e / %.3e of the scale (ours / library)" % (
        float(e_ours.max()) / scale, float(e_lib.max()) / scale, float(e_ours.pow(2).mean().sqrt()) / scale,
        float(e_lib.pow(2).mean().sqrt()) / scale)
    print(what, stats)
    assert float(e_ours.max()) <= float(e_lib.max()), stats
    assert float(e_ours.pow(2).mean().sqrt()) <= float(e_lib.pow(2).mean().sqrt()), stats


def test_accuracy_over_magnitudes_and_cancellation():
    """Rows scaled over 24 orders of magnitude (the split has no scale of its own), and sums that cancel to 1e-4 of their
    terms: the error stays relative to sum |a b|, as the fp32 GEMM's does."""
    torch.manual_seed(3)
    M, N, K = 1024, 128, 256
    a = torch.randn(M, K, device="cuda") * torch.logspace(-12, 12, M, device="cuda")[:, None]
    w = torch.randn(N, K, device="cuda")
    ref, absref = _ref64(a, w), a.double().abs() @ w.double().abs().t()
    ours = g3.gemm(a, g3.split_planes(w, False), g3.EPI_BIAS, bias=torch.zeros(N, device="cuda")).double()
    lib = (a @ w.t()).double()
    assert float(((ours - ref).abs() / absref).max()) <= float(((lib - ref).abs() / absref).max())
    assert float(((ours - ref).abs() / absref).max()) <= 2e-7


@pytest.mark.parametrize("M", [1, 127, 192, 1000])
def test_epilogues_and_ragged_rows(M):
    torch.manual_seed(4)
    N, K = 256, 64
    a, w = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    bias, aux = torch.randn(N, device="cuda"), torch.randn(M, N, device="cuda")
    planes = g3.split_planes(w, False)
    prod = (a.double() @ w.double().t())
    close = lambda x, y: float((x.double() - y).abs().max()) <= 2e-6 * max(1.0, float(y.abs().max()))
    assert close(g3.gemm(a, planes, g3.EPI_BIAS, bias=bias), prod + bias.double())
    assert close(g3.gemm(a, planes, g3.EPI_BIAS_RELU, bias=bias), (prod + bias.double()).relu())
    masked = g3.gemm(a, planes, g3.EPI_MASK, aux=aux)
    assert close(masked, torch.where(aux > 0, prod, torch.zeros_like(prod))) and bool((masked[aux <= 0] == 0).all())
    acc = aux.clone()
    out = g3.gemm(a, planes, g3.EPI_ADD, aux=acc, out=acc)      # in place: C += A B^T
    assert out.data_ptr() == acc.data_ptr() and close(acc, prod + aux.double())
    guard = torch.full((M + 8, N), 7.0, device="cuda")          # rows past M are never written
    g3.gemm(a, planes, g3.EPI_BIAS, bias=bias, out=guard[:M])
    assert bool((guard[M:] == 7.0).all())


def test_split_weight_follows_the_parameter_in_place():
    w = torch.nn.Parameter(torch.randn(128, 64, device="cuda"), requires_grad=False)
    sw = g3.SplitWeight(transpose=False)
    p0 = sw.planes(w)
    ptr = p0.data_ptr()
    assert sw.planes(w).data_ptr() == ptr
    before = p0.clone()
    with torch.no_grad():
        w.mul_(2.0)                                              # bumps _version
    p1 = sw.planes(w)
    assert p1.data_ptr() == ptr and not torch.equal(p1, before)  # same buffer, new contents
    a = torch.randn(64, 64, device="cuda")
    got = g3.gemm(a, p1, g3.EPI_BIAS, bias=torch.zeros(128, device="cuda"))
    assert float((got.double() - a.double() @ w.double().t()).abs().max()) <= 2e-5


def test_argument_errors_are_returned():
    a = torch.randn(8, 48, device="cuda")                        # K % 32 != 0
    with pytest.raises((RuntimeError, AssertionError)):
        g3.gemm(a, torch.zeros(3, 128, 48, device="cuda", dtype=torch.int16), g3.EPI_BIAS, bias=torch.zeros(128, device="cuda"))
    a = torch.randn(8, 64, device="cuda")
    with pytest.raises(RuntimeError):                            # epilogue without its operand
        g3.gemm(a, torch.zeros(3, 128, 64, device="cuda", dtype=torch.int16), g3.EPI_MASK)


def test_swin_linears_take_the_split_products_in_that_mode_and_follow_their_weights():
    """backbone._frozen_linear: under ``Switches.gemm_arith = "bf16x3"`` the frozen Swin linears with >= 8192 rows and fitting
    shapes go through the split-bf16 GEMM -- closer to fp64 than the library's fp32 GEMM --, smaller or unfitting ones stay on the
    library, and the cached planes follow an in-place weight change (Joiner.refresh_derived, as before a graph replay)."""
    from ziragroundingdino_amd import backbone, transformer as zt
    torch.manual_seed(0)
    mlp = backbone.Mlp(384, 1536).cuda()
    for p in mlp.parameters():
        p.requires_grad_(False)
    x = torch.randn(2, 4200, 384, device="cuda")
    ref64 = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(
        x.double(), mlp.fc1.weight.double(), mlp.fc1.bias.double())).float().double(), mlp.fc2.weight.double(), mlp.fc2.bias.double())
    old = zt.Switches.gemm_arith
    try:
        with torch.no_grad():
            zt.Switches.gemm_arith = "f32"
            lib = mlp(x)
            assert "_bf16x3_split" not in mlp.__dict__
            zt.Switches.gemm_arith = "bf16x3"
            got = mlp(x)
            assert set(k[0] for k in mlp.__dict__["_bf16x3_split"]) == {"fc1", "fc2"}
            small = mlp(x[:, :1000])                                   # 2000 rows: the library
            assert torch.equal(small, torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(
                x[:, :1000], mlp.fc1.weight, mlp.fc1.bias)), mlp.fc2.weight, mlp.fc2.bias))
            e_got, e_lib = (got.double() - ref64).abs().max().item(), (lib.double() - ref64).abs().max().item()
            assert e_got <= 1.5 * e_lib + 1e-7, (e_got, e_lib)
            # an in-place weight change is followed by the planes
            mlp.fc1.weight.mul_(0.5)
            j = backbone.Joiner(torch.nn.Sequential(mlp), torch.nn.Identity())
            j.refresh_derived()
            again = mlp(x)
            zt.Switches.gemm_arith = "f32"
            assert (again - mlp(x)).abs().max().item() <= 1e-4 * again.abs().max().item()
    finally:
        zt.Switches.gemm_arith = old
