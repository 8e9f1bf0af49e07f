#!/usr/bin/env python3
"""Dev: aten ops of one training step by launch count (torch.profiler), with input shapes and the Python frames
that issue them -- to find launch-count hot spots in the glue code."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
from ziragroundingdino_amd.config import zira_swint_config
from ziragroundingdino_amd.groundingdino import build_model
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch
dev = torch.device("cuda"); torch.manual_seed(0)
model = build_model(zira_swint_config(device="cuda")).to(dev).train()
trainer = ZiraTrainer(model)
data = synthetic_batch(2, 800, 1333, device=dev)
for _ in range(4):
    trainer.run_step(data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    trainer.run_step(data)
    torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU]
by = collections.defaultdict(lambda: [0, 0.0, 0])
for e in ev:
    nk = len(e.kernels)
    if nk == 0 or e.cpu_children and any(len(c.kernels) for c in e.cpu_children):
        continue                                    # leaf ops that launch kernels themselves
    stack = [f for f in (e.stack or []) if "ziragroundingdino_amd" in f or "bench" in f]
    where = stack[0].split("ziragroundingdino_amd/")[-1] if stack else "(autograd / other)"
    shapes = str([s for s in (e.input_shapes or []) if s])[:70]
    k = (e.name, shapes, where[:60])
    by[k][0] += nk
    by[k][1] += sum(kk.duration for kk in e.kernels)
print("leaf ops with kernels: %d launches" % sum(v[0] for v in by.values()))
tot = collections.defaultdict(lambda: [0, 0.0])
for (name, shapes, where), (n, t, _) in by.items():
    tot[where][0] += n; tot[where][1] += t
print("\nby source line (top 60 by launches):")
for where, (n, t) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:60]:
    print("%6d launches %9.1f us  %s" % (n, t, where))
print("\nby op / shape / line (top 170 by launches):")
for (name, shapes, where), (n, t, _) in sorted(by.items(), key=lambda kv: -kv[1][0])[:170]:
    print("%5d %8.1f us  %-28s %-70s %s" % (n, t, name[:28], shapes, where))
