#!/usr/bin/env python3
"""Capture the MSDA inputs of real model steps and time the kernels on them (device-bound, graph replay)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kbench import graphed, timeit
from ziragroundingdino_amd import _C
from ziragroundingdino_amd.config import zira_swint_config
from ziragroundingdino_amd.groundingdino import build_model
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch

dev = torch.device("cuda")
torch.manual_seed(0)
model = build_model(zira_swint_config()).to(dev).train()
trainer = ZiraTrainer(model)
data = synthetic_batch(2, 800, 1333, device=dev)
captured = {}
orig_b = _C.ms_deform_attn_backward
def hook(value, sh, st, loc, attn, go, step, **kw):
    key = "enc" if loc.shape[1] == value.shape[1] else "dec"
    captured.setdefault(key, []).append([t.detach().clone() for t in (value, sh, st, loc, attn, go)])
    return orig_b(value, sh, st, loc, attn, go, step, **kw)
cap_step = int(os.environ.get("ZIRA_CAPTURE_STEP", "2"))
for i in range(cap_step + 1):
    if i == cap_step:
        _C.ms_deform_attn_backward = hook
    trainer.run_step(data)
_C.ms_deform_attn_backward = orig_b
torch.cuda.synchronize()
if os.environ.get("ZIRA_SAVE_INPUTS"):
    out = {k: [t.cpu() for t in c[0]] for k, c in captured.items()}
    if os.environ.get("ZIRA_SAVE_ALL_DEC"):      # every decoder layer's call (backward order: last layer first)
        for i, c in enumerate(captured["dec"][1:], 1):
            out["dec%d" % i] = [t.cpu() for t in c]
    torch.save(out, os.environ["ZIRA_SAVE_INPUTS"])
    if os.environ.get("ZIRA_SAVE_ONLY"):
        sys.exit(0)
for key, calls in captured.items():
    for ci in (0, len(calls) - 1):
        v, sh, st, loc, attn, go = calls[ci]
        fwd = lambda: _C.ms_deform_attn_forward(v, sh, st, loc, attn, 64)
        bwd = lambda: orig_b(v, sh, st, loc, attn, go, 64)
        per = 10
        gf, gb = graphed(fwd, per), graphed(bwd, per)
        tf = min(timeit(gf, 3) / per for _ in range(3)); tb = min(timeit(gb, 3) / per for _ in range(3))
        l = loc.float()
        oob = ((l < 0) | (l > 1)).any(-1).float().mean().item()
        print("%s call %d: Q=%d fwd %.1f us bwd %.1f us | loc mean %.3f std %.3f, out-of-[0,1] samples %.1f%%, attn max %.3f"
              % (key, ci, loc.shape[1], tf, tb, l.mean().item(), l.std().item(), 100 * oob, attn.max().item()), flush=True)
        # pixel histogram at level 0: how concentrated are the samples?
        H, W = [int(x) for x in sh[0]]
        x = (l[:, :, :, 0, :, 0] * W).floor().clamp(0, W - 1).long(); y = (l[:, :, :, 0, :, 1] * H).floor().clamp(0, H - 1).long()
        idx = (y * W + x).flatten()
        cnt = torch.bincount(idx, minlength=H * W).float()
        print("   level-0 samples per pixel: max %d, mean %.2f, pixels hit %.1f%%" % (cnt.max().item(), cnt.mean().item(), 100 * (cnt > 0).float().mean().item()))

# entries per (b, m, level, tile) for T = 16 tiles per level (what the backward plan uses at Q=900)
v, sh, st, loc, attn, go = captured["dec"][0]
T = 16
B, Q, M, L, P, _ = loc.shape
for name, l_ in (("in-model", loc.float()), ("uniform", torch.rand_like(loc.float()))):
    counts = []
    for lvl in range(L):
        H, W = [int(x) for x in sh[lvl]]
        span = -(-(H * W) // T)
        x = l_[:, :, :, lvl, :, 0] * W - 0.5; y = l_[:, :, :, lvl, :, 1] * H - 0.5
        ok = (x > -1) & (y > -1) & (x < W) & (y < H)
        pix = (y.floor().clamp(0, H - 1) * W + x.floor().clamp(0, W - 1)).long()
        tile = (pix // span)                                            # [B,Q,M,P]
        key = (torch.arange(B, device=dev)[:, None, None, None] * M + torch.arange(M, device=dev)[None, None, :, None]) * T + tile
        c = torch.bincount(key[ok].flatten(), minlength=B * M * T).float() * 4
        counts.append(c)
        print("  %s level %d: entries/tile mean %.0f max %.0f  (max/mean %.1f)" % (name, lvl, c.mean().item(), c.max().item(), (c.max() / c.mean()).item()))
