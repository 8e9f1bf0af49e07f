#!/usr/bin/env python3
"""Soak of the hipGraph-replayed training step at the benchmark size: N steps (default 2000) over 4 rotating
minibatches with the front-end prefetch, every loss checked for finiteness, the matcher's device flags read at
the end.        python scripts/soak_graph.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd.config import zira_swint_config
from ziragroundingdino_amd.groundingdino import build_model
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device("cuda")
torch.manual_seed(0)
model = build_model(zira_swint_config(device="cuda")).to(dev).train()
model.use_transformer_graph = True
if os.environ.get("ZIRA_GRAPH_FUSION"):   # A/B: the fusion blocks replayed from graph pairs of their own (1) or launched eagerly (0)
    from ziragroundingdino_amd.graphs import GraphedTransformer
    GraphedTransformer.graph_fusion = os.environ["ZIRA_GRAPH_FUSION"] == "1"
trainer = ZiraTrainer(model)
batches = [synthetic_batch(2, 800, 1333, n_categories=15, seed=i, device=dev) for i in range(4)]
t0 = time.perf_counter()
bad = 0
for it in range(steps):
    out = trainer.run_step(batches[it % 4], next_data=batches[(it + 1) % 4])
    if it % 100 == 0 or it == steps - 1:
        tot = float(sum(out.values()))
        bad += int(tot != tot)
        print("step %5d  total loss %.4f  %.1f images/s" % (it, tot, 2 * (it + 1) / (time.perf_counter() - t0)), flush=True)
torch.cuda.synchronize()
model.criterion.matcher.check()
assert bad == 0
print("SOAK-OK %d steps, %.1f images/s" % (steps, 2 * steps / (time.perf_counter() - t0)))
