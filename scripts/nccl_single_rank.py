#!/usr/bin/env python3
"""RCCL sanity on one GPU: a 1-rank 'nccl' process group running the collectives the DP step
uses (flat side-branch gradient all-reduce, num_boxes all-reduce, barrier, MAX of the timer),
then two trainer steps with the trainer's all-reduce branch forced on."""
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.ones(4622853, device=dev)
dist.all_reduce(x); dist.barrier()
t = torch.tensor([1.5], device=dev, dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(x[0]) == 1.0 and float(t) == 1.5
from ziragroundingdino_amd.config import zira_swint_config
from ziragroundingdino_amd.groundingdino import build_model
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch
torch.manual_seed(0)
model = build_model(zira_swint_config(device=str(dev))).to(dev).train()
trainer = ZiraTrainer(model)
trainer.world = 2          # take the all-reduce + divide branch (with one rank: sum == own gradient)
ref = None
data = synthetic_batch(2, 800, 1333, seed=0, device=dev)
for i in range(2):
    out = trainer.run_step(data)
    print("step", i, {k: round(float(v), 4) for k, v in list(out.items())[:3]}, flush=True)
torch.cuda.synchronize()
dist.destroy_process_group()
print("nccl single-rank path OK")
