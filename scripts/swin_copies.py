import os, sys, torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, "/root/repo")
from ziragroundingdino_amd.config import zira_swint_config
from ziragroundingdino_amd.groundingdino import build_model
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch
from ziragroundingdino_amd.utils import nested_tensor_from_tensor_list
dev = torch.device("cuda"); torch.manual_seed(0)
model = build_model(zira_swint_config(device="cuda")).to(dev).train()
model.use_frontend_graphs = False
ZiraTrainer(model)   # (freezes what the task freezes)
data = synthetic_batch(2, 800, 1333, device=dev)
with torch.no_grad():
    samples = nested_tensor_from_tensor_list(model.preprocess_image(data))
    for _ in range(3):
        model.run_backbone(samples)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        model.run_backbone(samples)
        torch.cuda.synchronize()
import collections
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::_to_copy", "aten::fill_", "aten::zero_", "aten::bernoulli_", "aten::div_", "aten::empty"):
        chain, p = [], e.cpu_parent
        while p is not None and len(chain) < 4:
            chain.append(p.name[:40]); p = p.cpu_parent
        cnt[(e.name, str(e.input_shapes)[:60], " <- ".join(chain))] += 1
for k, n in cnt.most_common(25):
    print(n, k)
memc = collections.Counter(e.name for e in prof.events() if "Memcpy" in e.name or "Memset" in e.name)
print(memc)
