#!/usr/bin/env python3
"""Dev: which Python lines issue the device-to-device copies of one eager training step at the benchmark size
(aten::copy_ / clone / contiguous with device time, grouped by input shapes and the innermost package frame)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd.config import zira_swint_config  # noqa: E402
from ziragroundingdino_amd.groundingdino import build_model  # noqa: E402
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
model = build_model(zira_swint_config(device="cuda")).to(dev).train()
model.use_transformer_graph = False
trainer = ZiraTrainer(model)
batches = [synthetic_batch(2, 800, 1333, n_categories=15, seed=i, device=dev) for i in range(2)]
for i in range(3):
    trainer.run_step(batches[i % 2], next_data=batches[(i + 1) % 2])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    trainer.run_step(batches[1], next_data=batches[0])
    torch.cuda.synchronize()
BIG = 2 * 22223 * 256
seen = {}
for e in prof.events():
    if e.name != "aten::copy_" or not e.input_shapes or not e.input_shapes[0]:
        continue
    n = 1
    for d in e.input_shapes[0]:
        n *= d
    if n < int(os.environ.get("MIN_ELEMS", BIG // 8)):
        continue
    chain, p = [], e.cpu_parent
    while p is not None and len(chain) < 6:
        chain.append(p.name[:60])
        p = p.cpu_parent
    key = (str(e.input_shapes[0]), " <- ".join(chain))
    t = getattr(e, "device_time_total", 0)
    a = seen.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += t
print("large aten::copy_ calls of one eager step (shape, enclosing ops):")
print("total %.1f us in %d calls" % (sum(v[1] for v in seen.values()), sum(v[0] for v in seen.values())))
for (sh, ch), (n, t) in sorted(seen.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get("TOP", 60))]:
    print("%8.1f us x%-3d %-24s %s" % (t, n, sh, ch))
