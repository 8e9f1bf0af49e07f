"""The fused small attention (csrc/attn.hip, C ABI zira_attn_{fwd,bwd}_f32) against the composition it replaces
(scores = q k^T / sqrt(d) + mask, softmax, p v -- what nn.MultiheadAttention computes between its projections;
reference transformer_for_adapter.py:1029-1054), forward and all three gradients, fp32.  Tolerance 2e-5 of the tensor
scale for the forward, 1e-4 for the gradients (the sums over 900 keys are folded in another order)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from ziragroundingdino_amd import attention  # noqa: E402


def reference(q, k, v, H, key_mask):
    L, B, E = q.shape
    S, d = k.shape[0], E // H
    qh = q.reshape(L, B * H, d).transpose(0, 1)
    kh = k.reshape(S, B * H, d).transpose(0, 1)
    vh = v.reshape(S, B * H, d).transpose(0, 1)
    s = torch.bmm(qh, kh.transpose(1, 2)) / math.sqrt(d)
    if key_mask is not None:
        s = s + key_mask[:, None, None, :].expand(B, H, 1, S).reshape(B * H, 1, S)
    return torch.bmm(s.softmax(-1), vh).transpose(0, 1).reshape(L, B, E)


def close(a, b, tol, what):
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max()) / scale
    assert err <= tol, "%s: max err %.3e (scaled by %.3g) > %.1e" % (what, err, scale, tol)


@pytest.mark.parametrize("L,S,B,H,masked,strided", [
    (900, 900, 2, 8, False, True),     # decoder self-attention: q and k are slices of one fused projection
    (900, 32, 2, 8, True, True),       # decoder -> text cross-attention with a key-padding mask
    (37, 50, 3, 2, True, False),
    (64, 257, 1, 4, False, False),
    (5, 3, 2, 1, True, False),
    (301, 40, 1, 4, True, False),      # few key blocks, ragged tiles: dK / dV in two query shares + ordered sum
])
def test_fused_attention_matches_composition(L, S, B, H, masked, strided):
    g = torch.Generator().manual_seed(L * 131 + S)
    E = H * 32
    if strided:   # [rows, B, 2E] projections sliced along the last dimension
        qk = torch.randn(L, B, 2 * E, generator=g).cuda()
        kv = torch.randn(S, B, 2 * E, generator=g).cuda() if S != L else qk
        q = qk[..., :E].detach().requires_grad_()
        k = (kv[..., E:] if S != L else qk[..., E:]).detach().requires_grad_()
        v = torch.randn(S, B, E, generator=g).cuda().requires_grad_()
        q_in, k_in = qk[..., :E], (kv[..., E:] if S != L else qk[..., E:])
    else:
        q = torch.randn(L, B, E, generator=g).cuda().requires_grad_()
        k = torch.randn(S, B, E, generator=g).cuda().requires_grad_()
        v = torch.randn(S, B, E, generator=g).cuda().requires_grad_()
        q_in, k_in = q, k
    km = None
    if masked:
        km = torch.zeros(B, S)
        for b in range(B):
            km[b, S - 1 - (b % max(1, S - 1)):] = float("-inf")   # the last few keys of every batch element are padding
        km = km.cuda()
    go = torch.randn(L, B, E, generator=g).cuda()
    want = reference(q, k, v, H, km)
    gq, gk, gv = torch.autograd.grad(want, [q, k, v], go)

    assert attention.supported(q_in, k_in, v, H, km)
    q2 = q_in.detach().requires_grad_() if not strided else q_in.detach()
    # (strided inputs: gradients are taken with respect to fresh leaves that alias the same values)
    ql, kl, vl = (t.detach().clone().requires_grad_() for t in (q, k, v))
    got = attention.fused_attention(ql if not strided else _as_strided_like(ql, q_in), kl if not strided else _as_strided_like(kl, k_in),
                                    vl, H, km)
    close(got, want, 2e-5, "out")
    dq, dk, dv = torch.autograd.grad(got, [ql, kl, vl], go)
    close(dq, gq, 1e-4, "dq")
    close(dk, gk, 1e-4, "dk")
    close(dv, gv, 1e-4, "dv")


def _as_strided_like(leaf, view):
    """A view with the strides of ``view`` (a slice of a wider projection) holding the values of ``leaf``."""
    wide = torch.zeros(view.shape[0], view.shape[1], 2 * view.shape[2], device=leaf.device)
    off = 0 if view.storage_offset() % (2 * view.shape[2]) == 0 else view.shape[2]
    return _SliceCopy.apply(leaf, wide, off)


class _SliceCopy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, leaf, wide, off):
        E = leaf.shape[2]
        wide[..., off:off + E] = leaf
        return wide[..., off:off + E]

    @staticmethod
    def backward(ctx, g):
        return g, None, None


def test_fully_masked_query_rows_are_zero():
    q = torch.randn(40, 1, 32).cuda()
    k = torch.randn(8, 1, 32).cuda()
    km = torch.full((1, 8), float("-inf")).cuda()
    out = attention.fused_attention(q, k, k, 1, km)
    assert float(out.abs().max()) == 0.0
