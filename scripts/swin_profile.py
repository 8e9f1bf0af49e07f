#!/usr/bin/env python3
"""Dev: kernel time of the frozen Swin-T forward (eager, no grad) at the bench size, by kernel name (torch.profiler)."""
import os, sys, collections, re, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
from ziragroundingdino_amd.config import zira_swint_config
from ziragroundingdino_amd.groundingdino import build_model
dev = torch.device("cuda"); torch.manual_seed(0)
model = build_model(zira_swint_config(device="cuda")).to(dev).train()
x = torch.randn(2, 3, 800, 1333, device=dev)
mask = torch.zeros(2, 800, 1333, dtype=torch.bool, device=dev)
with torch.no_grad():
    for _ in range(3):
        model._backbone_tensors(x, mask)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes=True) as prof:
        model._backbone_tensors(x, mask)
        torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
ops = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        name = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", e.name)
        name = re.sub(r"<.*", "", name)[:50]
        agg[name][0] += 1; agg[name][1] += e.device_time if hasattr(e, "device_time") else e.cuda_time
    elif e.kernels and not any(c.kernels for c in e.cpu_children):
        k = (e.name, str([s for s in (e.input_shapes or []) if s])[:60])
        ops[k][0] += len(e.kernels); ops[k][1] += sum(kk.duration for kk in e.kernels)
tot = sum(v[1] for v in agg.values())
print("Swin-T + position embeddings forward: %d kernels, %.2f ms" % (sum(v[0] for v in agg.values()), tot / 1e3))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print("%8.1f us %5d x %7.2f  %s" % (t, n, t / n, k))
print("\nby op and shape:")
for (name, shp), (n, t) in sorted(ops.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%8.1f us %5d  %-26s %s" % (t, n, name[:26], shp))
