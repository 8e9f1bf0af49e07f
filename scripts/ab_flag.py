#!/usr/bin/env python3
"""Same-box A/B of a class-level switch: ms/step with the flag on, off, on, off."""
import os, sys, time, importlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd.config import zira_swint_config
from ziragroundingdino_amd.groundingdino import build_model
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch
modname, clsname, attr = sys.argv[1:4]
cls = getattr(importlib.import_module(modname), clsname)
dev = torch.device("cuda"); torch.manual_seed(0)
model = build_model(zira_swint_config()).to(dev).train(); trainer = ZiraTrainer(model)
data = synthetic_batch(2, 800, 1333, device=dev)
for _ in range(5): trainer.run_step(data)
for rep in range(3):
    for flag in (True, False):
        setattr(cls, attr, flag)
        for _ in range(3): trainer.run_step(data)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(15): trainer.run_step(data)
        torch.cuda.synchronize()
        print("%s.%s=%s: %.2f ms/step" % (clsname, attr, flag, (time.perf_counter() - t0) / 15 * 1e3), flush=True)
