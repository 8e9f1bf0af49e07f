#!/usr/bin/env python3
"""Regenerate ziragroundingdino_amd/tuned_gemm_gfx950.csv: run the BASELINE configs[1] training step
(and one eval forward) with PyTorch TunableOp tuning ON, so that every GEMM shape of the step gets
its fastest hipBLASLt / rocBLAS kernel recorded."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import tuned_gemm
out = sys.argv[1] if len(sys.argv) > 1 else tuned_gemm.DEFAULT_FILE
if os.path.exists(out) and not os.environ.get("ZIRA_TUNE_APPEND"):
    os.remove(out)
assert tuned_gemm.enable(out, tune=True)
from ziragroundingdino_amd.config import zira_swint_config
from ziragroundingdino_amd.groundingdino import build_model
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch
dev = torch.device("cuda"); torch.manual_seed(0)
model = build_model(zira_swint_config()).to(dev).train()
trainer = ZiraTrainer(model, tuned_gemms=False)
batches = [int(b) for b in os.environ.get("ZIRA_TUNE_BATCHES", "2,1,3,4").split(",")]
for bs in batches:                      # images per GPU: the BASELINE step first, then its neighbours
    data = synthetic_batch(bs, 800, 1333, device=dev)
    model.train()
    for i in range(3):
        trainer.run_step(data)
        torch.cuda.synchronize()
    model.eval()
    with torch.no_grad():
        model(data)
    torch.cuda.synchronize()
    print("batch", bs, "done:", len(torch.cuda.tunable.get_results()), "results", flush=True)
torch.cuda.tunable.write_file(out) if hasattr(torch.cuda.tunable, "write_file") else None
print("results:", len(torch.cuda.tunable.get_results()), "->", out)
