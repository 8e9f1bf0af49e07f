#!/usr/bin/env python3
"""Dev: the library GEMMs that remain in one eager training step, by ATen op, input shapes and the kernel that ran
(torch.profiler; `FRONT=1` with the frozen front end launched eagerly too)."""
import collections
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
model = build_model(zira_swint_config()).to(dev).train()
model.use_transformer_graph = False
model.use_frontend_graphs = os.environ.get("FRONT") != "1"
trainer = ZiraTrainer(model)
data = synthetic_batch(2, 800, 1333, n_categories=int(os.environ.get("CATEGORIES", "15")), device=dev)
for _ in range(3):
    trainer.run_step(data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    trainer.run_step(data)
    torch.cuda.synchronize()
rows = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.name in ("aten::mm", "aten::addmm", "aten::bmm", "aten::baddbmm"):
        for k in e.kernels:
            r = rows[(e.name, str(e.input_shapes)[:110], k.name[:60])]
            r[0] += 1
            r[1] += k.duration
tot = sum(r[1] for r in rows.values())
print("library GEMM kernels of one eager step: %.2f ms over %d launches" % (tot / 1e3, sum(r[0] for r in rows.values())))
for (op, shapes, kern), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print("%8.1f us x%-3d %-13s %-112s %s" % (us, n, op, shapes, kern))
