#!/usr/bin/env python3
"""Dev: the (N, a, b, transposed) of the tall reductions of one fusion block at the bench shape, and their times."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import dense, transformer  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
blk = transformer.BiAttentionBlock(v_dim=256, l_dim=256, embed_dim=1024, num_heads=4, dropout=0.0, drop_path=0.0).to(dev).train()
for p in blk.parameters():
    p.requires_grad_(False)
T = int(os.environ.get("TOKENS", 32))
v = torch.randn(2, 22223, 256, device=dev, requires_grad=True)
l = torch.randn(2, T, 256, device=dev, requires_grad=True)
mask_l = torch.zeros(2, T, dtype=torch.bool, device=dev)
seen = []
orig = dense.xty


def spy(X, Y, x_transposed=False):
    seen.append((tuple(X.shape), tuple(Y.shape), x_transposed))
    return orig(X, Y, x_transposed)


dense.xty = spy
ov, ol = blk(v, l, attention_mask_v=None, attention_mask_l=mask_l)
torch.autograd.grad([ov, ol], [v, l], [torch.randn_like(v), torch.randn_like(l)])
dense.xty = orig
for s in seen:
    X = torch.randn(s[0], device=dev)
    Y = torch.randn(s[1], device=dev)
    for _ in range(3):
        orig(X, Y, s[2])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        orig(X, Y, s[2])
    e1.record()
    torch.cuda.synchronize()
    print(s, "%.1f us (partial + fold)" % (e0.elapsed_time(e1) * 50))
