"""C = (A @ B) * (H > 0) (csrc/gemm_drelu.hip, C ABI zira_gemm_drelu_f32) against torch, and the frozen FFN that uses it in
its backward against the plain module chain (reference transformer_for_adapter.py:883-886: linear2(relu(linear1(x))))."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K", [(1800, 2048, 256), (44446, 2048, 256), (77, 128, 16), (129, 256, 48)])
def test_gemm_drelu_matches_torch(M, N, K):
    from ziragroundingdino_amd import _lib

    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).cuda()
    B = torch.randn(K, N, generator=g).cuda()
    H = torch.randn(M, N, generator=g).cuda().clamp_min(0)
    C = torch.full((M, N), float("nan"), device="cuda")
    rc = lib.zira_gemm_drelu_f32(A.data_ptr(), B.data_ptr(), H.data_ptr(), M, N, K, C.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    want = (A.double() @ B.double()) * (H > 0)
    err = (C.double() - want).abs().max().item()
    assert err <= 1e-5 * K ** 0.5 * 4, err          # fp32 accumulation over K terms of unit variance
    assert torch.equal(C == 0, (H <= 0) | (want == 0).bool())


def test_frozen_ffn_backward_matches_module_chain(monkeypatch):
    from ziragroundingdino_amd import transformer as T

    monkeypatch.setattr(T.Switches, "gemm_arith", "f32")   # (this is the library-fp32 form of the frozen FFN; the default is f16x2)
    torch.manual_seed(0)
    layer = T.DeformableTransformerEncoderLayer(256, 2048, 0.0, "relu", 4, 8, 4).cuda().train()
    for p in layer.parameters():
        p.requires_grad_(False)
    x = torch.randn(2, 5001, 256, device="cuda", requires_grad=True)
    go = torch.randn(2, 5001, 256, device="cuda")
    res = {}
    try:
        for flag in (False, True):
            T.Switches.fused_ffn_backward = flag
            out, _ = layer.forward_ffn(x)
            res[flag] = (out,) + torch.autograd.grad(out, [x], go)
    finally:
        T.Switches.fused_ffn_backward = True
    assert torch.equal(res[True][0], res[False][0])                      # the forward is the same two GEMMs
    torch.testing.assert_close(res[True][1], res[False][1], rtol=1e-4, atol=1e-4)
    # ... and so is the one-node form with the residual LayerNorm (what forward_ffn takes when everything is frozen)
    x2 = x.reshape(-1, 256)
    assert T._frozen_ffn_norm_ok(x2, layer.linear1, layer.linear2, layer.norm2)
    src2 = T._FrozenFFN.apply(x2, layer.linear1.weight, layer.linear1.bias, layer.linear2.weight, layer.linear2.bias)
    want = layer.norm2.add_norm(x, src2.view_as(x))
    torch.testing.assert_close(res[True][0], want, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(res[True][1], torch.autograd.grad(want, [x], go)[0], rtol=1e-4, atol=1e-4)
