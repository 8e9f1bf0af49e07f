#!/usr/bin/env python3
"""Dev: one image <-> text fusion attention (BiMultiHeadAttention, 22223 image tokens, 32 text tokens) forward + backward,
for `scripts/kstats_py.sh fusion scripts/fusion_target.py [iters]`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import transformer  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda")
torch.manual_seed(0)
att = transformer.BiMultiHeadAttention(v_dim=256, l_dim=256, embed_dim=1024, num_heads=4, dropout=0.0).to(dev)
for p in att.parameters():
    p.requires_grad_(False)
v = torch.randn(2, 22223, 256, device=dev, requires_grad=True)
l = torch.randn(2, 32, 256, device=dev, requires_grad=True)
mask_l = torch.zeros(2, 32, dtype=torch.bool, device=dev)
mask_l[1, 20:] = True
gv, gl = torch.randn_like(v), torch.randn_like(l)
for _ in range(iters):
    ov, ol = att(v, l, attention_mask_v=None, attention_mask_l=mask_l)
    torch.autograd.grad((ov * gv).sum() + (ol * gl).sum(), [v, l])
torch.cuda.synchronize()
