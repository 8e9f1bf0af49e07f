#!/usr/bin/env python3
"""Device time of one BiMultiHeadAttention forward+backward at the bench shape, by kernel."""
import os, sys, torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import transformer
dev = torch.device("cuda"); torch.manual_seed(0)
att = transformer.BiMultiHeadAttention(v_dim=256, l_dim=256, embed_dim=1024, num_heads=4, dropout=0.0).to(dev)
for p in att.parameters(): p.requires_grad_(False)
v = torch.randn(2, 22223, 256, device=dev, requires_grad=True)
l = torch.randn(2, 16, 256, device=dev, requires_grad=True)
mask_l = torch.zeros(2, 16, dtype=torch.bool, device=dev)
gv = torch.randn(2, 22223, 256, device=dev); gl = torch.randn(2, 16, 256, device=dev)
def step():
    ov, ol = att(v, l, attention_mask_v=None, attention_mask_l=mask_l)
    torch.autograd.grad((ov * gv).sum() + (ol * gl).sum(), [v, l])
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda r: -r.self_device_time_total)
tot = sum(r.self_device_time_total for r in rows)
print("total device time %.1f us over %d kernels" % (tot, sum(r.count for r in rows)))
for r in rows[:30]:
    print("  %8.1f us x%-3d %s" % (r.self_device_time_total, r.count, r.key[:110]))
