#!/usr/bin/env python3
"""GPU time per transformer sub-module (forward and backward), by synchronising hooks."""
import os, sys, time, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd.config import zira_swint_config
from ziragroundingdino_amd.groundingdino import build_model
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch

dev = torch.device("cuda")
torch.manual_seed(0)
model = build_model(zira_swint_config()).to(dev).train()
trainer = ZiraTrainer(model)
data = synthetic_batch(2, 800, 1333, device=dev)
for _ in range(3):
    trainer.run_step(data)
acc = collections.OrderedDict()
def sync():
    torch.cuda.synchronize(); return time.perf_counter()
def watch(mod, name):
    st = {}
    def pre(m, a): st["f"] = sync()
    def post(m, a, o): acc["fwd " + name] = acc.get("fwd " + name, 0.0) + sync() - st["f"]
    def bpre(m, g): st["b"] = sync()
    def bpost(m, gi, go): acc["bwd " + name] = acc.get("bwd " + name, 0.0) + sync() - st["b"]
    mod.register_forward_pre_hook(pre); mod.register_forward_hook(post)
    mod.register_full_backward_pre_hook(bpre); mod.register_full_backward_hook(bpost)
tr = model.transformer
for n, m in tr.encoder.named_children():
    print("encoder child:", n, type(m).__name__)
for n, m in tr.decoder.named_children():
    print("decoder child:", n, type(m).__name__)
enc = tr.encoder
for l in enc.layers:
    watch(l.self_attn, "enc.deform.self_attn(MSDA module)")
    watch(l, "enc.deform layer (attn+ffn)")
for l in enc.text_layers: watch(l, "enc.text layer")
for l in enc.fusion_layers: watch(l, "enc.fusion layer")
for l in tr.decoder.layers:
    watch(l, "dec layer")
    watch(l.cross_attn, "dec.cross_attn(MSDA module)")
N = 3
for _ in range(N):
    loss = sum(model(data).values()); loss.backward(); trainer.flat_grad.zero_()
for k, v in acc.items():
    print("%-44s %7.2f ms/step" % (k, v / N * 1e3))
