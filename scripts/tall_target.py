#!/usr/bin/env python3
"""Dev: the tall reductions P^T Q over the image tokens at the fusion block's shapes: csrc/xty_bf16x3.hip (three bfloat16 planes on
the bf16 matrix cores) beside csrc/xty.hip (fp32 matrix instruction) and torch.bmm: us per call and error against fp64."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import dense  # noqa: E402

torch.manual_seed(0)
B, N = 2, 22223


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for n in (64, 128, 388, 776):
    for thin_first in (True, False):
        X = torch.rand(B, N, n if thin_first else 256, device="cuda")
        Y = torch.randn(B, N, 256 if thin_first else n, device="cuda")
        ref = torch.bmm(X.double().transpose(1, 2), Y.double())
        res = {}
        for name, flag in (("bf16x3", True), ("fp32 mfma", False)):
            dense.USE_TALL_BF16X3, dense.TALL_BF16X3_MIN_COLS = flag, 4
            out = dense.xty(X, Y)
            res[name] = (timed(lambda: dense.xty(X, Y)), float((out.double() - ref).abs().max() / ref.abs().max()))
        dense.USE_TALL_BF16X3, dense.TALL_BF16X3_MIN_COLS = True, 192
        t_lib = timed(lambda: torch.bmm(X.transpose(1, 2), Y))
        e_lib = float((torch.bmm(X.transpose(1, 2), Y).double() - ref).abs().max() / ref.abs().max())
        print("[%d,%d]^T [%d,%d]:  bf16x3 %6.1f us (err %.1e)   fp32 mfma %6.1f us (err %.1e)   torch.bmm %6.1f us (err %.1e)" % (
            N, X.shape[2], N, Y.shape[2], res["bf16x3"][0], res["bf16x3"][1], res["fp32 mfma"][0], res["fp32 mfma"][1], t_lib, e_lib))
