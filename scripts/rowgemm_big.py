#!/usr/bin/env python3
"""Dev: the row GEMMs (csrc/rowgemm.hip) at the ENCODER's row count (2 x 22223 image tokens) beside the library ops."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd.rowgemm import rowgemm  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
M = int(os.environ.get("ROWS", 44446))
x = torch.randn(M, 256, device=dev)
pos = torch.randn(M, 256, device=dev)
res = torch.randn(M, 256, device=dev)
gam, bet = torch.ones(256, device=dev), torch.zeros(256, device=dev)
wt = {n: (torch.randn(256, n, device=dev) / 16) for n in (256, 384, 2048)}   # [K, N]
w = {n: wt[n].t().contiguous() for n in wt}                                   # [N, K]
b = {n: torch.randn(n, device=dev) for n in wt}
x384 = torch.randn(M, 384, device=dev)
mean = x.mean(-1)
rstd = (x.var(-1, unbiased=False) + 1e-5).rsqrt()


def timeit(fn, name):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(5):
            fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    print("%-44s %8.2f us" % (name, e0.elapsed_time(e1) * 1e3 / 20), flush=True)


# check once
ref = torch.addmm(b[256], x, wt[256])
got = rowgemm(x, wt[256], w_is_nk=False, bias=b[256])
print("max diff", (ref - got).abs().max().item())
timeit(lambda: torch.addmm(b[256], x, wt[256]), "lib addmm N=256")
timeit(lambda: rowgemm(x, wt[256], w_is_nk=False, bias=b[256]), "rowgemm bias N=256")
timeit(lambda: torch.addmm(b[384], x + pos, wt[384]), "lib add + addmm N=384")
timeit(lambda: rowgemm(x, wt[384], w_is_nk=False, bias=b[384], pos=pos), "rowgemm pos+bias N=384")
timeit(lambda: F.layer_norm(torch.addmm(b[256], x, wt[256]) + res, (256,), gam, bet), "lib addmm + add + LN")
timeit(lambda: rowgemm(x, wt[256], w_is_nk=False, bias=b[256], res=res, ln=(gam, bet, 1e-5), ln_save=True), "rowgemm bias+res+LN")
timeit(lambda: res.clone().addmm_(x, wt[256]), "lib clone + addmm_ (beta=1)")
timeit(lambda: rowgemm(x, wt[256], w_is_nk=False, res=res), "rowgemm res N=256")
timeit(lambda: rowgemm(x384, w[384], w_is_nk=False, res=res), "rowgemm res K=384 N=256")
timeit(lambda: res.clone().addmm_(x384, w[384]), "lib clone + addmm_ K=384")
timeit(lambda: rowgemm(x, wt[256], w_is_nk=False, lnb=(x, gam, mean, rstd), lnb_save=True), "rowgemm LNbwd + dgrad N=256")
timeit(lambda: x.clone(), "clone (for scale)")
# the fusion block's image-side output: v_n + gamma * (P R + b) per image, K = H * T = 128
M1 = 22223
P = torch.rand(2, M1, 128, device=dev)
R = torch.randn(2, 128, 256, device=dev) / 11
vn = torch.randn(2, M1, 256, device=dev)
gamma = torch.rand(256, device=dev)
bo = torch.randn(256, device=dev)


def lib_path():
    d = torch.empty_like(vn)
    for i in range(2):
        torch.addmm(bo, P[i], R[i], out=d[i])
    return torch.addcmul(vn, d, gamma)


def fused_path():
    Rs = R * gamma
    bs = bo * gamma
    out = torch.empty_like(vn)
    for i in range(2):
        rowgemm(P[i], Rs[i], w_is_nk=False, bias=bs, res=vn[i], out=out[i])
    return out


print("max diff", (lib_path() - fused_path()).abs().max().item())
timeit(lib_path, "fusion out_v: 2 addmm + addcmul")
timeit(fused_path, "fusion out_v: 2 rowgemm (gamma folded, res)")
