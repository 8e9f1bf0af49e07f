#!/usr/bin/env python3
"""Wall-clock breakdown of the training step with synchronisation at section boundaries."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd.config import zira_swint_config
from ziragroundingdino_amd.groundingdino import build_model
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch
from ziragroundingdino_amd.utils import nested_tensor_from_tensor_list

dev = torch.device("cuda")
torch.manual_seed(0)
model = build_model(zira_swint_config()).to(dev).train()
trainer = ZiraTrainer(model)
data = synthetic_batch(2, 800, 1333, device=dev)
for _ in range(3):
    trainer.run_step(data)

def sync():
    torch.cuda.synchronize(); return time.perf_counter()

acc = {}
def add(k, dt): acc[k] = acc.get(k, 0.0) + dt
N = 5
crit = model.criterion
orig_forward = crit.forward
def timed_crit(*a, **k):
    t0 = sync(); r = orig_forward(*a, **k); add("criterion(7 matchings+losses)", sync() - t0); return r
crit.forward = timed_crit
orig_tr = model.transformer.forward
def timed_tr(*a, **k):
    t0 = sync(); r = orig_tr(*a, **k); add("transformer fwd (host enqueue only)", time.perf_counter() - t0)
    add("transformer fwd", sync() - t0); return r
model.transformer.forward = timed_tr
orig_bb = model.run_backbone
def timed_bb(*a, **k):
    t0 = sync(); r = orig_bb(*a, **k); add("backbone fwd (host enqueue only)", time.perf_counter() - t0)
    add("backbone fwd", sync() - t0); return r
model.run_backbone = timed_bb
orig_txt = model.encode_text
def timed_txt(*a, **k):
    t0 = sync(); r = orig_txt(*a, **k); add("text (tokenize+masks+bert+rsb)", sync() - t0); return r
model.encode_text = timed_txt
for _ in range(N):
    t0 = sync()
    loss_dict = model(data)
    t1 = sync(); add("forward total", t1 - t0)
    losses = sum(loss_dict.values()); losses.backward()
    add("backward (host enqueue only)", time.perf_counter() - t1)
    t2 = sync(); add("backward", t2 - t1)
    total_norm = torch.linalg.vector_norm(trainer.flat_grad, 2)
    trainer.flat_grad.mul_(torch.clamp(0.1 / (total_norm + 1e-6), max=1.0))
    trainer.optimizer.step(); trainer.flat_grad.zero_()
    t3 = sync(); add("clip+adamw", t3 - t2)
for k, v in acc.items():
    print("%-34s %7.2f ms" % (k, v / N * 1e3))
