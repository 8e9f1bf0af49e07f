#!/usr/bin/env python3
"""Dev: device time of one fusion block (BiAttentionBlock: LayerNorms, bi-directional attention, layer-scale residuals) forward +
backward at the bench shape (22223 image tokens x 2 images, 32 text tokens), by kernel."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import transformer  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
blk = transformer.BiAttentionBlock(v_dim=256, l_dim=256, embed_dim=1024, num_heads=4, dropout=0.0, drop_path=0.0).to(dev).train()
for p in blk.parameters():
    p.requires_grad_(False)
v = torch.randn(2, 22223, 256, device=dev, requires_grad=True)
l = torch.randn(2, 32, 256, device=dev, requires_grad=True)
mask_l = torch.zeros(2, 32, dtype=torch.bool, device=dev)
gv, gl = torch.randn_like(v), torch.randn_like(l)


def step():
    ov, ol = blk(v, l, attention_mask_v=None, attention_mask_l=mask_l)
    torch.autograd.grad([ov, ol], [v, l], [gv, gl])


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
rows = sorted((r for r in prof.key_averages() if r.self_device_time_total > 0), key=lambda r: -r.self_device_time_total)
tot = sum(r.self_device_time_total for r in rows)
print("total device time %.1f us over %d kernels" % (tot, sum(r.count for r in rows)))
for r in rows[:50]:
    print("  %8.1f us x%-3d %s" % (r.self_device_time_total, r.count, r.key[:130]))
