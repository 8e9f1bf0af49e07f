#!/usr/bin/env python3
"""Device time of one encoder deformable layer forward+backward at the bench shape, by kernel (weights frozen)."""
import os, sys, torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import transformer
from bench import NORTH_STAR_SHAPES
dev = torch.device("cuda"); torch.manual_seed(0)
layer = transformer.DeformableTransformerEncoderLayer(256, 2048, 0.0, "relu", 4, 8, 4).to(dev).train()
for p in layer.parameters(): p.requires_grad_(False)
S = sum(h * w for h, w in NORTH_STAR_SHAPES); B = 2
src = torch.randn(B, S, 256, device=dev, requires_grad=True)
pos = torch.randn(B, S, 256, device=dev)
shapes = torch.tensor(NORTH_STAR_SHAPES, device=dev)
start = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
ref = torch.rand(B, S, 4, 2, device=dev) * 0.8 + 0.1
g = torch.randn(B, S, 256, device=dev)
def step():
    out = layer(src=src, pos=pos, reference_points=ref, spatial_shapes=shapes, level_start_index=start, key_padding_mask=None)
    out = out[0] if isinstance(out, tuple) else out
    torch.autograd.grad((out * g).sum(), [src])
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda r: -r.self_device_time_total)
rows = [r for r in rows if r.self_device_time_total > 0]
tot = sum(r.self_device_time_total for r in rows)
print("total device time %.1f us over %d kernels" % (tot, sum(r.count for r in rows)))
for r in rows[:40]:
    print("  %8.1f us x%-3d %s" % (r.self_device_time_total, r.count, r.key[:130]))
