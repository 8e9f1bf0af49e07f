#!/usr/bin/env python3
"""Host-side timing of the stacked criterion on the bench shapes (7 sets x 2 images x 900 queries)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
from ziragroundingdino_amd import criterion

dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
S, B, Q, C = 7, 2, 900, 256
logits = (torch.randn(S, B, Q, C, generator=g)).to(dev).requires_grad_(True)
boxes = torch.cat([torch.rand(S, B, Q, 2, generator=g) * 0.6 + 0.2, torch.rand(S, B, Q, 2, generator=g) * 0.3 + 0.05], -1).to(dev).requires_grad_(True)
targets = [{"labels": torch.randint(0, 7, (5,), generator=g).to(dev),
            "boxes": torch.cat([torch.rand(5, 2, generator=g) * 0.5 + 0.25, torch.rand(5, 2, generator=g) * 0.3 + 0.1], -1).to(dev)} for _ in range(B)]
crit = criterion.build_criterion(SimpleNamespace(aux_loss=True, dec_layers=6, max_text_len=C)).to(dev)
suffixes = ["_%d" % i for i in range(5)] + ["", "_enc"]
out_loop = {"pred_logits": logits[5], "pred_boxes": boxes[5],
            "aux_outputs": [{"pred_logits": logits[i], "pred_boxes": boxes[i]} for i in range(5)],
            "enc_outputs": {"pred_logits": logits[6], "pred_boxes": boxes[6]}}
out_fast = dict(out_loop, stacked=(logits, boxes, suffixes))

def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

print("loop    criterion fwd      %.2f ms" % t(lambda: crit(out_loop, targets)))
print("stacked criterion fwd      %.2f ms" % t(lambda: crit(out_fast, targets)))
print("stacked matcher only       %.2f ms" % t(lambda: crit.matcher.forward_stacked(logits.detach(), boxes.detach(), targets)))
def cost_only():
    flat = {"pred_logits": logits.detach().reshape(1, S * B * Q, -1), "pred_boxes": boxes.detach().reshape(1, S * B * Q, 4)}
    return crit.matcher.cost_matrix(flat, targets, check=False)
print("cost matrix (device only)  %.2f ms" % t(cost_only))
def fb():
    l = crit(out_fast, targets); sum(l[k] * crit.weight_dict[k] for k in l).backward()
print("stacked fwd+bwd            %.2f ms" % t(fb))
def fbl():
    l = crit(out_loop, targets); sum(l[k] * crit.weight_dict[k] for k in l).backward()
print("loop    fwd+bwd            %.2f ms" % t(fbl))
