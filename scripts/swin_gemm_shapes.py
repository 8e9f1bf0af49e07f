#!/usr/bin/env python3
"""Dev: the library's fp32 GEMM (with the committed TunableOp winners) at every linear of Swin-T on 2 x 800 x 1333 images: us,
TF/s -- which shapes the tuned file serves badly."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import tuned_gemm  # noqa: E402

print("tuned file enabled:", tuned_gemm.enable())
rows = {1: 2 * 200 * 334, 2: 2 * 100 * 167, 3: 2 * 50 * 84, 4: 2 * 25 * 42}
pad7 = lambda h, w: 2 * (-(-h // 7) * 7) * (-(-w // 7) * 7)
C = {1: 96, 2: 192, 3: 384, 4: 768}
n = {1: 2, 2: 2, 3: 6, 4: 2}
tot = 0.0
for st in (1, 2, 3, 4):
    c, M = C[st], rows[st]
    for name, K, N in (("qkv", c, 3 * c), ("proj", c, c), ("fc1", c, 4 * c), ("fc2", 4 * c, c)):
        R = int(os.environ.get("ROTATE", "1"))     # distinct operand sets visited in turn (cold operands when their sum exceeds the caches)
        xs = [torch.randn(M, K, device="cuda") for _ in range(R)]
        ws = [torch.randn(N, K, device="cuda") * 0.05 for _ in range(R)]
        outs = [torch.empty(M, N, device="cuda") for _ in range(R)]
        b = torch.randn(N, device="cuda")
        for i in range(5):
            torch.addmm(b, xs[i % R], ws[i % R].t(), out=outs[i % R])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20):
            torch.addmm(b, xs[i % R], ws[i % R].t(), out=outs[i % R])
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e3
        tot += t * n[st]
        print("stage %d %-4s [%6d,%4d]x[%4d,%4d]  %7.1f us  %6.1f TF/s   x%d blocks" % (st, name, M, K, K, N, t, 2.0 * M * K * N / t / 1e6, n[st]))
print("sum over the 12 blocks: %.2f ms" % (tot / 1e3))
