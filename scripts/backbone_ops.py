#!/usr/bin/env python3
"""torch.profiler over the frozen Swin-T forward alone (eager, no graph): device time by op."""
import os, sys
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd.config import zira_swint_config
from ziragroundingdino_amd.groundingdino import build_model
from ziragroundingdino_amd.train import synthetic_batch
from ziragroundingdino_amd.utils import nested_tensor_from_tensor_list
dev = torch.device("cuda")
torch.manual_seed(0)
model = build_model(zira_swint_config()).to(dev).train()
model.before_train()
model.use_frontend_graphs = False
data = synthetic_batch(2, 800, 1333, device=dev)
samples = nested_tensor_from_tensor_list(model.preprocess_image(data))
for _ in range(3):
    model.run_backbone(samples)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    model.run_backbone(samples)
    torch.cuda.synchronize()
flat = sorted(prof.key_averages(), key=lambda r: -r.self_device_time_total)
print("== backbone: top ops by device time")
for r in flat[:28]:
    print("   %8.2f ms x%-5d %s" % (r.self_device_time_total / 1e3, r.count, r.key[:100]))
rows = prof.key_averages(group_by_input_shape=True)
for name in ["aten::copy_", "aten::addmm", "aten::native_layer_norm", "aten::add", "aten::_softmax", "aten::bmm", "aten::gelu"]:
    sel = sorted([r for r in rows if r.key == name], key=lambda r: -r.self_device_time_total)
    print("== %s: %.2f ms, %d calls" % (name, sum(r.self_device_time_total for r in sel) / 1e3, sum(r.count for r in sel)))
    for r in sel[:5]:
        print("   %8.1f us x%-4d %s" % (r.self_device_time_total, r.count, str(r.input_shapes)[:140]))
