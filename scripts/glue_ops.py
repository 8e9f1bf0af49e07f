#!/usr/bin/env python3
"""Dev: the ATen elementwise launches (copy / add / fill / mul / ...) of ONE eager training step at the benchmark size, counted
by (op, shape, innermost frame of this package, enclosing ATen / autograd ops) -- which lines issue the small launches the
hand-written nodes have not absorbed.  `python scripts/glue_ops.py [copy_|add|fill_|...] ` (substring of the op name; default: all
ops whose own device time is an elementwise kernel)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd.config import zira_swint_config  # noqa: E402
from ziragroundingdino_amd.groundingdino import build_model  # noqa: E402
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch  # noqa: E402

want = sys.argv[1] if len(sys.argv) > 1 else ""
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
FAMILY = ("aten::copy_", "aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::mul", "aten::mul_", "aten::sub", "aten::div",
          "aten::div_", "aten::neg", "aten::where", "aten::masked_fill_", "aten::sigmoid", "aten::clamp", "aten::cat", "aten::sum",
          "aten::native_layer_norm", "aten::native_layer_norm_backward", "aten::gelu", "aten::gelu_backward", "aten::index_select",
          "aten::gather", "aten::index", "aten::exp", "aten::log", "aten::sqrt", "aten::rsqrt", "aten::pow", "aten::addcmul_",
          "aten::addcdiv_", "aten::lerp_", "aten::threshold_backward", "aten::relu", "aten::_softmax", "aten::_softmax_backward_data")
seen = {}
for e in prof.events():
    if e.name not in FAMILY or (want and want not in e.name):
        continue
    t = getattr(e, "self_device_time_total", 0) or 0
    if t <= 0:
        continue
    frame = ""
    for s in (e.stack or []):
        if "ziragroundingdino_amd" in s:
            frame = s.split("ziragroundingdino_amd/")[-1][:70]
            break
    chain, p = [], e.cpu_parent
    while p is not None and len(chain) < 3:
        chain.append(p.name[:40])
        p = p.cpu_parent
    shape = str(e.input_shapes[0] if e.input_shapes else "")
    key = (e.name, shape, frame, " <- ".join(chain))
    a = seen.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += t
by_op = {}
for (name, _, _, _), (n, t) in seen.items():
    a = by_op.setdefault(name, [0, 0.0])
    a[0] += n
    a[1] += t
print("ATen elementwise-family ops with device time, one eager step: %d launches, %.1f us" % (
    sum(v[0] for v in by_op.values()), sum(v[1] for v in by_op.values())))
for name, (n, t) in sorted(by_op.items(), key=lambda kv: -kv[1][1]):
    print("  %-36s x%-5d %9.1f us" % (name, n, t))
print()
for (name, sh, fr, ch), (n, t) in sorted(seen.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get("TOP", 120))]:
    print("x%-4d %8.1f us  %-22s %-26s %-62s %s" % (n, t, name, sh[:26], fr, ch))
