#!/usr/bin/env python3
"""Dev: one frozen fusion block forward + backward at the benchmark size (2 x 22223 image tokens, 2 x 32 text tokens), text side
native (NATIVE=1, default) or on ATen ops (NATIVE=0); run under scripts/kstats_py.sh for kernel times."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import transformer as zt  # noqa: E402

torch.manual_seed(0)
blk = zt.BiAttentionBlock(v_dim=256, l_dim=256, embed_dim=1024, num_heads=4, dropout=0.0, drop_path=0.1).cuda().train()
for p in blk.parameters():
    p.requires_grad_(False)
blk.native_text_side = os.environ.get("NATIVE", "1") == "1"
T = int(os.environ.get("T", 32))
v = torch.randn(2, 22223, 256, device="cuda", requires_grad=True)
l = torch.randn(2, T, 256, device="cuda", requires_grad=True)
mask_l = torch.zeros(2, T, dtype=torch.bool, device="cuda")
for _ in range(int(os.environ.get("ITERS", 20))):
    ov, ol = blk(v, l, attention_mask_v=None, attention_mask_l=mask_l)
    torch.autograd.grad(ov.sum() + ol.sum(), [v, l])
torch.cuda.synchronize()
