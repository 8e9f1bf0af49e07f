#!/usr/bin/env python3
"""Dev: the small elementwise ATen ops of one eager training step (copy_, mul, add, add_, div, cat, sum, fill_ ...) grouped by
(op, input shape, enclosing ops): where the launch-sized kernels between the GEMMs come from.  `python scripts/small_ops.py [op]`"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd.config import zira_swint_config  # noqa: E402
from ziragroundingdino_amd.groundingdino import build_model  # noqa: E402
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch  # noqa: E402

want = sys.argv[1] if len(sys.argv) > 1 else None
OPS = {"aten::copy_", "aten::mul", "aten::add", "aten::add_", "aten::div", "aten::cat", "aten::sum", "aten::fill_", "aten::sub",
       "aten::neg", "aten::bitwise_not", "aten::where", "aten::div_", "aten::mul_", "aten::clamp_min", "aten::masked_fill_"}
dev = torch.device("cuda")
torch.manual_seed(0)
model = build_model(zira_swint_config(device="cuda")).to(dev).train()
model.use_transformer_graph = False
trainer = ZiraTrainer(model)
batches = [synthetic_batch(2, 800, 1333, n_categories=15, seed=i, device=dev) for i in range(2)]
for i in range(3):
    trainer.run_step(batches[i % 2], next_data=batches[(i + 1) % 2])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    trainer.run_step(batches[1], next_data=batches[0])
    torch.cuda.synchronize()
seen = {}
for e in prof.events():
    if e.name not in OPS or (want and want not in e.name):
        continue
    t = getattr(e, "device_time_total", 0)
    if t <= 0:
        continue
    chain, p = [], e.cpu_parent
    while p is not None and len(chain) < 4:
        chain.append(p.name[:48])
        p = p.cpu_parent
    key = (e.name, str(e.input_shapes[:2])[:60], " <- ".join(chain))
    a = seen.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += t
tot = sum(v[1] for v in seen.values())
print("%.2f ms over %d calls" % (tot / 1e3, sum(v[0] for v in seen.values())))
for (op, sh, ch), (n, t) in sorted(seen.items(), key=lambda kv: -kv[1][1])[:70]:
    print("%8.1f us x%-3d %-12s %-60s %s" % (t, n, op[6:], sh, ch))
