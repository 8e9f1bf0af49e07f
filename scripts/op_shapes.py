#!/usr/bin/env python3
"""torch.profiler over one training step: device time by (op, input shapes) for the elementwise
families that dominate the non-GEMM kernel time."""
import os, sys
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd.config import zira_swint_config
from ziragroundingdino_amd.groundingdino import build_model
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch
dev = torch.device("cuda")
torch.manual_seed(0)
model = build_model(zira_swint_config()).to(dev).train()
model.use_transformer_graph = False
model.use_frontend_graphs = bool(int(os.environ.get("FRONT_GRAPH", "0")))
trainer = ZiraTrainer(model)
data = synthetic_batch(2, 800, 1333, device=dev)
for _ in range(3):
    trainer.run_step(data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    trainer.run_step(data)
    torch.cuda.synchronize()
rows = prof.key_averages(group_by_input_shape=True)
flat = sorted(prof.key_averages(), key=lambda r: -r.self_device_time_total)
tot_all = sum(r.self_device_time_total for r in flat)
print("== top ops by device time (total %.1f ms)" % (tot_all / 1e3))
for r in flat[:32]:
    print("   %8.2f ms x%-5d %s" % (r.self_device_time_total / 1e3, r.count, r.key[:90]))
want = sys.argv[1:] or ["aten::copy_", "aten::add", "aten::add_", "aten::native_layer_norm", "aten::clamp", "aten::mul", "aten::fill_", "aten::threshold_backward", "aten::masked_fill"]
for name in want:
    sel = sorted([r for r in rows if r.key == name], key=lambda r: -r.self_device_time_total)
    tot = sum(r.self_device_time_total for r in sel)
    print("== %s: %.2f ms device, %d calls" % (name, tot / 1e3, sum(r.count for r in sel)))
    for r in sel[:int(os.environ.get("TOP", "8"))]:
        print("   %8.1f us x%-4d %s" % (r.self_device_time_total, r.count, str(r.input_shapes)[:150]))
