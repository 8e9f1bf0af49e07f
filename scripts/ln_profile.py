#!/usr/bin/env python3
"""Row LayerNorm kernel vs ATen at the shapes of a step (graph replay of 20 calls, us per call, TB/s)."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kbench import graphed, timeit
from ziragroundingdino_amd import dense
dev = "cuda"
for rows, C in [(44446, 256), (133600, 96), (33400, 192), (8400, 384), (2100, 768), (133600, 384)]:
    x = torch.randn(rows, C, device=dev); w = torch.randn(C, device=dev); b = torch.randn(C, device=dev)
    res = []
    for fn in (lambda: F.layer_norm(x, (C,), w, b), lambda: dense._LayerNorm.apply(x, w, b, 1e-5)):
        g = graphed(fn, 20)
        res.append(min(timeit(g, 3) / 20 for _ in range(3)))
    gb = 2 * rows * C * 4 / 1e3
    print("rows %6d C %4d: aten %7.1f us (%.2f TB/s)  kernel %7.1f us (%.2f TB/s)" % (rows, C, res[0], gb / res[0] / 1e3, res[1], gb / res[1] / 1e3), flush=True)
