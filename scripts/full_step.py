#!/usr/bin/env python3
"""Full GroundingDINO-T + ZiRa training steps on synthetic 800x1333 batches (developer run)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd.config import zira_swint_config
from ziragroundingdino_amd.groundingdino import build_model
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
H, W = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (800, 1333)
dev = torch.device("cuda")
torch.manual_seed(0)
t0 = time.time()
model = build_model(zira_swint_config()).to(dev).train()
print("params %.1f M, build %.1f s" % (sum(p.numel() for p in model.parameters()) / 1e6, time.time() - t0))
trainer = ZiraTrainer(model)
model.use_transformer_graph = os.environ.get("ZIRA_GRAPH", "1") == "1"
print("trainable %d tensors, %d values" % (len(trainer.params), trainer.flat_grad.numel()))
data = synthetic_batch(bs, H, W, device=dev)
for i in range(steps):
    torch.cuda.synchronize(); t = time.time()
    out = trainer.run_step(data)
    torch.cuda.synchronize(); dt = time.time() - t
    tot = float(sum(out.values()))
    print("step %d: %.1f ms  total loss %.4f  mem %.1f GB" % (i, dt * 1e3, tot, torch.cuda.max_memory_allocated() / 2**30), flush=True)
print({k: round(float(v), 4) for k, v in out.items()})
