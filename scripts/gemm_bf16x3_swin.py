#!/usr/bin/env python3
"""Dev: the split-bf16 GEMM beside the library's fp32 GEMM at the Swin-T linears' shapes that fit it (N % 128 == 0, K % 32 == 0),
800 x 1333 images, batch 2: us per call, replayed back to back."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import gemm_bf16x3 as g3  # noqa: E402
from ziragroundingdino_amd import tuned_gemm  # noqa: E402

tuned_gemm.enable()
shapes = [("s1 fc1", 133600, 96, 384), ("s2 fc1", 33400, 192, 768), ("s3 qkv", 8400, 384, 1152), ("s3 proj", 8400, 384, 384),
          ("s3 fc1", 8400, 384, 1536), ("s3 fc2", 8400, 1536, 384), ("s4 qkv", 2100, 768, 2304), ("s4 proj", 2100, 768, 768),
          ("s4 fc1", 2100, 768, 3072), ("s4 fc2", 2100, 3072, 768)]


def bench(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, M, K, N in shapes:
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    planes = g3.split_planes(w, False)
    t_lib = bench(lambda: torch.addmm(b, x, w.t()))
    t_g3 = bench(lambda: g3.gemm(x, planes, g3.EPI_BIAS, bias=b))
    fl = 2.0 * M * K * N
    print("%-8s [%6d,%4d]x[%4d,%4d]  library %7.1f us (%5.1f TF/s)  bf16x3 %7.1f us (%5.1f TF/s)  x%.2f" % (
        name, M, K, K, N, t_lib, fl / t_lib / 1e6, t_g3, fl / t_g3 / 1e6, t_lib / t_g3))
