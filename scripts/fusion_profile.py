#!/usr/bin/env python3
"""Dev: kernels of the six decoder layers (forward + backward, eager) at the bench size: inputs captured from a real step,
then the decoder alone under torch.profiler; kernels by total time and by op / shape."""
import os, sys, collections, re, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
from ziragroundingdino_amd.config import zira_swint_config
from ziragroundingdino_amd.groundingdino import build_model
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch
dev = torch.device("cuda"); torch.manual_seed(0)
model = build_model(zira_swint_config(device="cuda")).to(dev).train()
trainer = ZiraTrainer(model)
data = synthetic_batch(2, 800, 1333, device=dev)
cap = {}
dec = model.transformer.encoder.fusion_layers[0]
h = dec.register_forward_pre_hook(lambda m, a, k: cap.update(args=a, kwargs=k), with_kwargs=True)
trainer.run_step(data); h.remove()
def det(x):
    if torch.is_tensor(x):
        y = x.detach().clone()
        return y.requires_grad_(x.requires_grad and x.is_floating_point())
    return x
args = [det(a) for a in cap["args"]]; kwargs = {k: det(v) for k, v in cap["kwargs"].items()}
def run():
    out = dec(*args, **kwargs)
    loss = sum(x.float().sum() for x in out)
    loss.backward()
for _ in range(2): run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], record_shapes=True) as prof:
    run(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0]); ops = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        name = re.sub(r"\(anonymous namespace\)::|void |at::native::", "", e.name); name = re.sub(r"<.*", "", name)[:48]
        agg[name][0] += 1; agg[name][1] += e.device_time if hasattr(e, "device_time") else e.cuda_time
    elif e.kernels and not any(c.kernels for c in e.cpu_children):
        k = (e.name, str([s for s in (e.input_shapes or []) if s])[:64]); ops[k][0] += len(e.kernels); ops[k][1] += sum(kk.duration for kk in e.kernels)
tot = sum(v[1] for v in agg.values())
print("one fusion block fwd+bwd: %d kernels, %.2f ms" % (sum(v[0] for v in agg.values()), tot / 1e3))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print("%8.1f us %5d x %7.2f  %s" % (t, n, t / n, k))
print("\nby op and shape:")
for (name, shp), (n, t) in sorted(ops.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%8.1f us %5d  %-30s %s" % (t, n, name[:30], shp))
