#!/usr/bin/env python3
"""Full-size model, a few batch shapes: one training step each must run and give finite losses."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd.config import zira_swint_config
from ziragroundingdino_amd.groundingdino import build_model
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch
dev = torch.device("cuda")
torch.manual_seed(0)
model = build_model(zira_swint_config()).to(dev).train()
trainer = ZiraTrainer(model)
for (B, H, W) in [(1, 800, 1333), (3, 800, 1333), (4, 640, 901), (2, 512, 777), (5, 800, 1333)]:
    data = synthetic_batch(B, H, W, seed=B, device=dev)
    t0 = time.perf_counter()
    for _ in range(2):
        out = trainer.run_step(data)
    torch.cuda.synchronize()
    ok = all(bool(torch.isfinite(v)) for v in out.values())
    print("B=%d %dx%d: finite=%s, %.1f ms/step (incl. first-call setup), peak mem %.1f GB" % (
        B, H, W, ok, (time.perf_counter() - t0) / 2 * 1e3, torch.cuda.max_memory_allocated() / 1e9), flush=True)
    assert ok
print("sweep OK")
