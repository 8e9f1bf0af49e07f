#!/usr/bin/env python3
"""Dev: ATen ops of one eager training step at the benchmark size by device time, grouped by (op, input shapes) --
which copies / adds / fills touch the 45 MB tensors.  `python scripts/op_profile.py [copy|add|all]`"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd.config import zira_swint_config  # noqa: E402
from ziragroundingdino_amd.groundingdino import build_model  # noqa: E402
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "all"
if os.environ.get("GEMM_ARITH"):      # f32 | bf16x3
    from ziragroundingdino_amd import transformer as _zt
    _zt.Switches.gemm_arith = os.environ["GEMM_ARITH"]
dev = torch.device("cuda")
torch.manual_seed(0)
model = build_model(zira_swint_config(device="cuda")).to(dev).train()
model.use_transformer_graph = False
trainer = ZiraTrainer(model)
data = synthetic_batch(2, 800, 1333, n_categories=15, device=dev)
for _ in range(3):
    trainer.run_step(data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    trainer.run_step(data)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", None)
    if t is None:
        t = getattr(e, "self_cuda_time_total", 0)
    if t <= 0:
        continue
    if what != "all" and what not in e.key:
        continue
    rows.append((t, e.count, e.key, str(e.input_shapes)[:150]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("device time of the listed ops: %.2f ms" % (tot / 1e3))
for t, n, k, sh in rows[:70]:
    print("%9.1f us  x%-4d %-44s %s" % (t, n, k[:44], sh))
