"""Tall-reduction product kernel (csrc/xty.hip, C ABI zira_xty_f32) against torch.bmm in float64,
and the two autograd wrappers BiMultiHeadAttention uses, against plain torch autograd."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from ziragroundingdino_amd import dense  # noqa: E402

DEV = "cuda"


def _rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("B,N,a,b,xt", [(2, 22223, 64, 256, False), (2, 22223, 64, 256, True),
                                        (2, 22223, 256, 64, False), (1, 5000, 36, 256, True),
                                        (3, 4097, 200, 132, False), (2, 2500, 4, 8, True),
                                        (2, 3000, 320, 260, False),
                                        # the fusion block's shapes at 4 heads x 32 tokens: the 16-byte-load form (xty_rows128)
                                        (2, 22223, 128, 256, False), (2, 22223, 256, 128, False), (1, 4096, 128, 256, False),
                                        (3, 4101, 256, 128, False), (1, 8191, 128, 256, False)])
def test_xty_matches_float64(B, N, a, b, xt):
    g = torch.Generator().manual_seed(N + a + b)
    X = torch.randn((B, a, N) if xt else (B, N, a), generator=g).to(DEV)
    Y = torch.randn(B, N, b, generator=g).to(DEV)
    assert dense._use_xty(X, Y, a, b, N)
    got = dense.xty(X, Y, x_transposed=xt)
    want = torch.bmm((X if xt else X.transpose(1, 2)).double(), Y.double())
    assert got.shape == (B, a, b)
    assert _rel(got, want) < 2e-6
    assert torch.equal(got, dense.xty(X, Y, x_transposed=xt))            # fixed fold order: bit-stable


def test_autograd_wrappers_match_torch():
    g = torch.Generator().manual_seed(3)
    B, N, a, d = 2, 9000, 48, 256
    P = torch.randn(B, a, N, generator=g).to(DEV).requires_grad_(True)
    V = torch.randn(B, N, d, generator=g).to(DEV).requires_grad_(True)
    go = torch.randn(B, a, d, generator=g).to(DEV)
    got = torch.autograd.grad((dense.tall_reduce(P, V) * go).sum(), [P, V])
    want = torch.autograd.grad((torch.bmm(P, V) * go).sum(), [P, V])
    for x, y in zip(got, want):
        assert _rel(x, y.double()) < 1e-5
    L = torch.randn(B, N, a, generator=g).to(DEV).requires_grad_(True)
    R = torch.randn(B, a, d, generator=g).to(DEV).requires_grad_(True)
    bias = torch.randn(d, generator=g).to(DEV).requires_grad_(True)
    go = torch.randn(B, N, d, generator=g).to(DEV)
    got = torch.autograd.grad((dense.wide_matmul(L, R, bias) * go).sum(), [L, R, bias])
    want = torch.autograd.grad((torch.baddbmm(bias, L, R) * go).sum(), [L, R, bias])
    for x, y in zip(got, want):
        assert _rel(x, y.double()) < 1e-5


@pytest.mark.parametrize("T,with_masks", [(16, True), (9, False), (32, True), (50, True), (256, True), (194, True), (70, False),
                                          (100, True), (161, True), (208, False)])   # (T > 64: a wave per image token)
def test_fused_bi_softmax_matches_unfused_attention(T, with_masks):
    """BiMultiHeadAttention with the fused score post-processing (csrc/bisoftmax.hip) against the
    same module running the PyTorch chain (itself pinned to the reference's order of operations and
    golden vectors): outputs and all gradients, with text / image padding masks."""
    from ziragroundingdino_amd import transformer

    torch.manual_seed(T)
    att = transformer.BiMultiHeadAttention(v_dim=256, l_dim=256, embed_dim=1024, num_heads=4, dropout=0.0).to(DEV)
    for p in att.parameters():
        p.data.normal_(0, 0.05)
    N = 3001
    v = torch.randn(2, N, 256, device=DEV, requires_grad=True)
    l = torch.randn(2, T, 256, device=DEV, requires_grad=True)
    mask_v = mask_l = None
    if with_masks:
        mask_v = torch.zeros(2, N, dtype=torch.bool, device=DEV)
        mask_v[1, 2500:] = True
        mask_l = torch.zeros(2, T, dtype=torch.bool, device=DEV)
        mask_l[0, T - 3:] = True
    gv, gl = torch.randn(2, N, 256, device=DEV), torch.randn(2, T, 256, device=DEV)
    res = {}
    for flag in (False, True):
        att.fused_softmax = flag
        ov, ol = att(v, l, attention_mask_v=mask_v, attention_mask_l=mask_l)
        grads = torch.autograd.grad((ov * gv).sum() + (ol * gl).sum(), [v, l] + list(att.parameters()))
        res[flag] = (ov, ol) + grads
    names = ["out_v", "out_l", "grad_v", "grad_l"] + ["grad " + n for n, _ in att.named_parameters()]
    for n, a, b in zip(names, res[True], res[False]):
        err = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
        assert err < 1e-4, (n, err)    # fp32 re-association over ~3000-term sums


@pytest.mark.parametrize("drop_path", [0.0, 0.4])
def test_fusion_block_residual_in_gemm_matches_addcmul(drop_path):
    """v + drop_path(gamma_v * delta_v) inside the attention's last GEMM (dense._FusionImageSide, the node that holds the block's
    image side since round 6 -- layer scale folded into the small operands, residual rows added by the GEMM's epilogue -- and
    dense._WideMatmulResidual with that node switched off) against the addcmul pass behind the GEMM: same RNG draws, values and
    gradients to fp32 re-association."""
    from ziragroundingdino_amd import transformer
    torch.manual_seed(0)
    blk = transformer.BiAttentionBlock(v_dim=256, l_dim=256, embed_dim=1024, num_heads=4, dropout=0.0, drop_path=drop_path,
                                       init_values=0.2).to(DEV).train()
    for p in blk.parameters():
        p.requires_grad_(False)
    v = torch.randn(2, 5003, 256, device=DEV, requires_grad=True)
    l = torch.randn(2, 32, 256, device=DEV, requires_grad=True)
    mask_l = torch.zeros(2, 32, dtype=torch.bool, device=DEV)
    mask_l[1, 29:] = True
    gv, gl = torch.randn_like(v), torch.randn_like(l)
    res = {}
    calls = []
    real, real_node = dense._WideMatmulResidual.forward, dense._FusionImageSide.forward

    def spy(ctx, *a):
        calls.append("gemm")
        return real(ctx, *a)

    def spy_node(ctx, *a):
        calls.append("node")
        return real_node(ctx, *a)

    dense._WideMatmulResidual.forward = staticmethod(spy)
    dense._FusionImageSide.forward = staticmethod(spy_node)
    try:
        for flag in ("node", "gemm", False):
            transformer.BiAttentionBlock.residual_in_gemm = bool(flag)
            transformer.BiAttentionBlock.fused_image_side = flag == "node"
            torch.manual_seed(11)
            ov, ol = blk(v, l, attention_mask_v=None, attention_mask_l=mask_l)
            res[flag] = (ov.detach(), ol.detach()) + torch.autograd.grad([ov, ol], [v, l], [gv, gl])
    finally:
        transformer.BiAttentionBlock.residual_in_gemm = transformer.BiAttentionBlock.fused_image_side = True
        dense._WideMatmulResidual.forward = staticmethod(real)
        dense._FusionImageSide.forward = staticmethod(real_node)
    assert calls == ["node", "gemm"]                         # each GEMM form ran when it was switched on, and only then
    for flag in ("node", "gemm"):
        for a, b in zip(res[flag], res[False]):
            assert _rel(a, b.double()) < 2e-6


def test_composed_text_side_matches_two_step_projections():
    """The text side's double projections through embed_dim (l_proj then the query weights, values_l_proj then the output
    weights, the value weights then out_l_proj) as one constant matrix each while the six Linears are frozen
    (BiMultiHeadAttention._composed_text_side) against the two-step form: outputs and input gradients to fp32
    re-association; a weight changed in place is followed, in the same buffers (captured graphs read them)."""
    from ziragroundingdino_amd import transformer
    torch.manual_seed(5)
    att = transformer.BiMultiHeadAttention(v_dim=256, l_dim=256, embed_dim=1024, num_heads=4, dropout=0.0).to(DEV)
    for p in att.parameters():
        p.data.normal_(0, 0.05)
        p.requires_grad_(False)
    N, T = 3001, 32
    v = torch.randn(2, N, 256, device=DEV, requires_grad=True)
    l = torch.randn(2, T, 256, device=DEV, requires_grad=True)
    mask_l = torch.zeros(2, T, dtype=torch.bool, device=DEV)
    mask_l[0, T - 3:] = True
    gv, gl = torch.randn(2, N, 256, device=DEV), torch.randn(2, T, 256, device=DEV)

    def run(flag):
        transformer.BiMultiHeadAttention.compose_text_side = flag
        try:
            ov, ol = att(v, l, attention_mask_v=None, attention_mask_l=mask_l)
            return (ov.detach(), ol.detach()) + torch.autograd.grad((ov * gv).sum() + (ol * gl).sum(), [v, l])
        finally:
            transformer.BiMultiHeadAttention.compose_text_side = True

    def check():
        for n, a, b in zip(("out_v", "out_l", "grad_v", "grad_l"), run(True), run(False)):
            assert _rel(a, b.double()) < 2e-5, n

    assert att._composed_text_side(l) is not None
    check()
    ptrs = [t.data_ptr() for t in att._text_side[1]]
    with torch.no_grad():
        att.l_proj.weight.mul_(1.5)
        att.out_l_proj.bias.add_(0.25)
    check()
    assert [t.data_ptr() for t in att._text_side[1]] == ptrs
    att.l_proj.weight.requires_grad_(True)      # a trainable Linear: the two-step form
    assert att._composed_text_side(l) is None


@pytest.mark.parametrize("B,N,n,thin_first", [(2, 22223, 776, True), (2, 22223, 776, False), (1, 5000, 196, True), (3, 4101, 388, False),
                                              (2, 2500, 1024, True), (2, 3333, 36, True), (1, 40, 256, False), (2, 22223, 128, True)])
def test_tall_reduction_on_the_bf16_matrix_cores(B, N, n, thin_first):
    """csrc/xty_bf16x3.hip (three bfloat16 planes, six product terms, fp32 sums) at COCO-length captions (776 = 4 heads x 194
    tokens), ragged widths and row counts, both output orientations: against fp64 no worse than 1.5e-6 of the largest element
    (the fp32 matrix-instruction kernel of csrc/xty.hip: 3e-7; torch.bmm: 3e-6), bit-stable, beside the fp32 kernel."""
    g = torch.Generator().manual_seed(N + n)
    X = torch.randn((B, N, n if thin_first else 256), generator=g).to(DEV)
    Y = torch.randn((B, N, 256 if thin_first else n), generator=g).to(DEV)
    if thin_first:
        X = X.softmax(1) * 50.0                      # (probabilities over the tokens: a few large, most tiny)
    want = torch.bmm(X.double().transpose(1, 2), Y.double())
    got = dense._tall_bf16x3(X, Y)
    assert got.shape == want.shape
    assert _rel(got, want) < 1.5e-6
    assert torch.equal(got, dense._tall_bf16x3(X, Y))
    if N >= 2048:
        fp32 = dense._xty_native(X, Y, False)
        assert _rel(got, fp32.double()) < 2e-6
    # the dispatcher takes it from 192 columns (under the split arithmetics)
    assert dense._use_tall_bf16x3(X, Y, X.shape[2], Y.shape[2], N, False) == (n >= 192)
