#!/usr/bin/env python3
"""Dev: the thin products of the fusion block's image side (csrc/thin_f16x2.hip) beside the library's bmm and the row GEMM they
replace, at the benchmark's shapes: us per call (events around 50 calls after 10), for `scripts/kstats_py.sh` too."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import dense  # noqa: E402
from ziragroundingdino_amd.rowgemm import rowgemm  # noqa: E402

torch.manual_seed(0)
B, M = 2, 22223


def timed(fn, n=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for n in (64, 128):
    v = torch.randn(B, M, 256, device="cuda")
    a = torch.randn(B, 256, n, device="cuda")
    e = torch.rand(B, M, n, device="cuda")
    e2 = torch.rand(B, M, n, device="cuda")
    z = torch.randn(B, n, 256, device="cuda")
    z2 = torch.randn(B, n, 256, device="cuda")
    bias = torch.randn(B, 256, device="cuda")
    out = torch.empty(B, M, 256, device="cuda")
    print("H T = %d" % n)
    print("  [M,256] x [256,%d]:  thin %6.1f us   library bmm %6.1f us" % (n, timed(lambda: dense.thin_bmm(v, a, True)), timed(lambda: torch.bmm(v, a))))
    print("  [M,%d] x [%d,256]:  thin %6.1f us   library bmm %6.1f us" % (n, n, timed(lambda: dense.thin_bmm(e, z, True)), timed(lambda: torch.bmm(e, z))))

    def rg():
        if n < 128:                      # (the row GEMM needs K >= 128: the module falls back to addmm per image + addcmul)
            for i in range(B):
                torch.addmm(bias[i], e[i], z[i], out=out[i])
            return torch.addcmul(v, out, bias[0])
        for i in range(B):
            rowgemm(e[i], z[i], w_is_nk=False, bias=bias[i], res=v[i], out=out[i])
    print("  ... + bias + residual: thin %6.1f us   row GEMM per image / addmm + addcmul %6.1f us" % (
        timed(lambda: dense.thin_bmm(e, z, True, bias=bias, res=v, out=out)), timed(rg)))
    print("  two sources in one pass: thin %6.1f us   two library bmm + add %6.1f us" % (
        timed(lambda: dense.thin_bmm(e, z, True, A2=e2, W2=z2)), timed(lambda: torch.bmm(e, z) + torch.bmm(e2, z2))))
