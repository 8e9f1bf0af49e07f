#!/usr/bin/env python3
"""Dev: kernel time of the tall frozen linears of the encoder (44446 x K -> N) by call path."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import tuned_gemm
from bench import timeit

dev = torch.device("cuda")
M = int(os.environ.get("M", 44446))
for tuned in (False, True):
    if tuned:
        tuned_gemm.enable()
    for K, N in ((256, 256), (256, 128), (256, 384), (256, 2048), (2048, 256)):
        x = torch.randn(2, M // 2, K, device=dev)
        W = torch.randn(N, K, device=dev) * 0.05
        b = torch.randn(N, device=dev)
        Wt = W.t().contiguous()
        # a second large tensor pass between calls so that x is not served from a warm cache
        spoil = torch.empty(1 << 27, device=dev)
        paths = {
            "F.linear(x,W,b)": lambda: F.linear(x, W, b),
            "addmm(b,x2d,Wt)": lambda: torch.addmm(b, x.view(-1, K), Wt),
            "x2d@Wt": lambda: x.view(-1, K) @ Wt,
            "F.linear(x,W)": lambda: F.linear(x, W),
            "dgrad g@W": lambda: x.view(-1, K) @ W.t().contiguous().t() if False else (torch.empty(0)),
        }
        g = torch.randn(M, N, device=dev)
        paths["dgrad g@W"] = lambda: g @ W
        for name, fn in paths.items():
            for _ in range(3):
                fn()
            t_warm = timeit(fn, 20)
            def cold():
                spoil.zero_()
                fn()
            t_z = timeit(lambda: spoil.zero_(), 10)
            t_cold = timeit(cold, 10) - t_z
            print("tuned=%d K=%4d N=%4d %-18s warm %7.1f us   after 512MB fill %7.1f us" % (tuned, K, N, name, t_warm, t_cold), flush=True)
