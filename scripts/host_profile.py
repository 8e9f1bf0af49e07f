#!/usr/bin/env python3
"""cProfile of the host side of a few training steps (where does the Python/launch time go?)."""
import cProfile, pstats, os, sys, io, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd.config import zira_swint_config
from ziragroundingdino_amd.groundingdino import build_model
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch
dev = torch.device("cuda"); torch.manual_seed(0)
model = build_model(zira_swint_config()).to(dev).train(); trainer = ZiraTrainer(model)
data = synthetic_batch(2, 800, 1333, device=dev)
for _ in range(5): trainer.run_step(data)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): trainer.run_step(data)
torch.cuda.synchronize(); pr.disable()
for key in ("tottime", "cumtime"):
    s = io.StringIO(); pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(int(os.environ.get("TOP", "28")))
    print(s.getvalue()[:12000])
