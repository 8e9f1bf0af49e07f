#!/usr/bin/env python3
"""Dev: device time of one text-enhancer layer (transformer.TransformerEncoderLayer: self-attention under the block-diagonal
sub-sentence mask + FFN, post-LN) forward + backward at the bench shape (32 text tokens x 2 images), by kernel."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import transformer  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
T = int(os.environ.get("TOKENS", 32))
lay = transformer.TransformerEncoderLayer(d_model=256, nhead=4, dim_feedforward=1024, dropout=0.0).to(dev).train()
for p in lay.parameters():
    p.requires_grad_(False)
src = torch.randn(T, 2, 256, device=dev, requires_grad=True)
pos = torch.randn(T, 2, 256, device=dev)
mask = torch.ones(2, T, T, dtype=torch.bool, device=dev)
for i in range(0, T, 4):
    mask[:, i:i + 4, i:i + 4] = False          # True = not allowed (nn.MultiheadAttention convention)
g = torch.randn_like(src)


def step():
    out = lay(src, src_mask=mask, src_key_padding_mask=None, pos=pos)
    torch.autograd.grad([out], [src], [g])


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
rows = sorted((r for r in prof.key_averages() if r.self_device_time_total > 0), key=lambda r: -r.self_device_time_total)
tot = sum(r.self_device_time_total for r in rows)
print("total device time %.1f us over %d kernels" % (tot, sum(r.count for r in rows)))
for r in rows[:40]:
    print("  %8.1f us x%-3d %s" % (r.self_device_time_total, r.count, r.key[:130]))
