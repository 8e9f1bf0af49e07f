#!/usr/bin/env python3
"""Dev: the panel GEMM (csrc/gemm_f16x2_panel.hip) and the thin products (csrc/thin_f16x2.hip) at the encoder's row count,
30 calls each, for scripts/pmc_py.sh / kstats_py.sh."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import dense, gemm_bf16x3 as g3  # noqa: E402

torch.manual_seed(0)
M = 44446
a = torch.randn(M, 256, device="cuda")
b = torch.randn(256, device="cuda")
for N in (256, 384):
    w = torch.randn(N, 256, device="cuda") * 0.05
    pf = g3.split_frags_f16x2(w, False)
    out = torch.empty(M, N, device="cuda")
    for _ in range(30):
        g3.gemm_f16x2_panel(a, pf, N, g3.EPI_BIAS, bias=b.repeat(2)[:N].contiguous(), out=out)
v = a.view(2, M // 2, 256)
for n in (128,):
    e = torch.rand(2, M // 2, n, device="cuda")
    for _ in range(30):
        dense.thin_bmm(v, torch.randn(2, 256, n, device="cuda"), True)
        dense.thin_bmm(e, torch.randn(2, n, 256, device="cuda"), True)
torch.cuda.synchronize()
