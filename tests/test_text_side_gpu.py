"""The text side of a frozen fusion block on csrc/textside.hip (text_side.py; BiAttentionBlock.native_text_side) against the
same block on ATen ops (``native_text_side = False``: the path tests/test_modules_golden.py pins to the reference's
BiAttentionBlock golden vectors, fuse_modules.py:252-305): both outputs and both input gradients, with padding on both
sides, several text lengths, and stochastic depth drawn from the same generator state."""
import pytest
import torch

from ziragroundingdino_amd import text_side, transformer as zt

pytestmark = pytest.mark.gpu


def _block(drop_path, seed):
    torch.manual_seed(seed)
    blk = zt.BiAttentionBlock(v_dim=256, l_dim=256, embed_dim=1024, num_heads=4, dropout=0.0, drop_path=drop_path).cuda()
    for n, p in blk.named_parameters():
        if p.dim() > 1:
            p.data.normal_(0, 0.05)
        elif "gamma" in n:
            p.data.uniform_(0.5, 1.5)
        else:
            p.data.normal_(0, 0.1)
        if "layer_norm" in n and n.endswith("weight"):
            p.data.add_(1.0)
        p.requires_grad_(False)
    return blk.train()


@pytest.mark.parametrize("case", [(2, 1500, 32, 0.0), (2, 1500, 32, 0.3), (3, 700, 9, 0.0), (1, 400, 195, 0.0)])
def test_native_text_side_matches_the_aten_block(case):
    B, N, T, dp = case
    blk = _block(dp, seed=T)
    g = torch.Generator().manual_seed(B * 10 + T)
    v0 = torch.randn(B, N, 256, generator=g).cuda()
    l0 = torch.randn(B, T, 256, generator=g).cuda()
    mask_v = torch.zeros(B, N, dtype=torch.bool, device="cuda")
    mask_v[-1, N - N // 5:] = True
    mask_l = torch.zeros(B, T, dtype=torch.bool, device="cuda")
    mask_l[0, T - max(1, T // 4):] = True
    gv, gl = torch.randn(B, N, 256, generator=g).cuda(), torch.randn(B, T, 256, generator=g).cuda()
    res = {}
    for native in (False, True):
        blk.native_text_side = native
        used = []
        orig = text_side.text_out
        text_side.text_out = lambda *a, **k: (used.append(1), orig(*a, **k))[1]
        try:
            torch.manual_seed(123)
            v, l = v0.clone().requires_grad_(True), l0.clone().requires_grad_(True)
            ov, ol = blk(v, l, attention_mask_v=mask_v, attention_mask_l=mask_l)
            grads = torch.autograd.grad((ov * gv).sum() + (ol * gl).sum(), [v, l])
        finally:
            text_side.text_out = orig
        assert bool(used) == native
        res[native] = (ov.detach(), ol.detach()) + grads
    for name, a, b in zip(("out_v", "out_l", "grad_v", "grad_l"), res[True], res[False]):
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() <= 2e-5 * scale, (name, (a - b).abs().max().item(), scale)


def test_native_text_side_declines_trainable_projections():
    blk = _block(0.0, seed=1)
    blk.attn.l_proj.weight.requires_grad_(True)
    v = torch.randn(1, 300, 256, device="cuda")
    l = torch.randn(1, 8, 256, device="cuda")
    assert blk._forward_native_text(blk.layer_norm_v(v), l, None, None) is None
    blk.attn.l_proj.weight.requires_grad_(False)
    blk.gamma_l.requires_grad_(True)
    assert blk._forward_native_text(blk.layer_norm_v(v), l, None, None) is None
