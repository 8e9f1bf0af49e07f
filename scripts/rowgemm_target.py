#!/usr/bin/env python3
"""Dev: the decoder's row GEMMs (csrc/rowgemm.hip) beside the library ops they replace, at M = 1800; run under
scripts/kstats_py.sh for kernel times."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd.rowgemm import rowgemm  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
M = 1800
x = torch.randn(M, 256, device=dev)
pos = torch.randn(M, 256, device=dev)
res = torch.randn(M, 256, device=dev)
gam, bet = torch.ones(256, device=dev), torch.zeros(256, device=dev)
ws = {n: torch.randn(n, 256, device=dev) / 16 for n in (256, 384, 512, 768, 2048)}
w2048 = torch.randn(256, 2048, device=dev) / 45
h = torch.randn(M, 2048, device=dev).relu()
g768 = torch.randn(M, 768, device=dev)
mean = x.mean(-1)
rstd = (x.var(-1, unbiased=False) + 1e-5).rsqrt()
for it in range(5):
    rowgemm(x, ws[768], w_is_nk=True, pos=pos, pos_cols=512)                       # NK2 N=768
    rowgemm(x, ws[256], w_is_nk=True, res=res, ln=(gam, bet, 1e-5), ln_save=True)   # NK4 LN K=256
    rowgemm(h, w2048, w_is_nk=True, res=res, ln=(gam, bet, 1e-5), ln_save=True)     # NK4 LN K=2048
    rowgemm(x, ws[2048], w_is_nk=True, relu=True)                                   # NK2 N=2048
    rowgemm(g768, ws[768], w_is_nk=False, res=res)                                  # KN2 K=768
    rowgemm(x, ws[256], w_is_nk=False, lnb=(x, gam, mean, rstd), lnb_save=True)     # KN2 lnb N=256
    rowgemm(x, w2048, w_is_nk=False, mask=h, lnb=(x, gam, mean, rstd), lnb_save=True)  # KN2 lnb mask N=2048
    rowgemm(h, ws[2048], w_is_nk=False, res=res)                                    # KN2 K=2048
    # what they replace
    F.linear(x + pos, ws[768])
    F.layer_norm(F.linear(x, ws[256]) + res, (256,), gam, bet)
    F.layer_norm(F.linear(h, w2048) + res, (256,), gam, bet)
    F.linear(x, ws[2048]).relu()
    torch.addmm(res, g768, ws[768])
    torch.addmm(res, h, ws[2048])
torch.cuda.synchronize()
# event timing of each rowgemm form (graph replay of 20 calls)
def timeit(fn, name):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(20):
            fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    print("%-34s %7.2f us" % (name, e0.elapsed_time(e1) * 1e3 / 100), flush=True)
timeit(lambda: rowgemm(x, ws[768], w_is_nk=True, pos=pos, pos_cols=512), "qkv N=768 K=256 (+pos)")
timeit(lambda: F.linear(x + pos, ws[768]), "  lib: add + linear")
timeit(lambda: rowgemm(x, ws[256], w_is_nk=True, res=res, ln=(gam, bet, 1e-5), ln_save=True), "out+res+LN K=256")
timeit(lambda: F.layer_norm(F.linear(x, ws[256]) + res, (256,), gam, bet), "  lib: linear + add + LN")
timeit(lambda: rowgemm(h, w2048, w_is_nk=True, res=res, ln=(gam, bet, 1e-5), ln_save=True), "linear2+res+LN K=2048")
timeit(lambda: F.layer_norm(F.linear(h, w2048) + res, (256,), gam, bet), "  lib: linear + add + LN")
timeit(lambda: rowgemm(x, ws[2048], w_is_nk=True, relu=True), "linear1+relu N=2048")
timeit(lambda: F.linear(x, ws[2048]).relu(), "  lib: linear + relu")
timeit(lambda: rowgemm(g768, ws[768], w_is_nk=False, res=res), "dgrad K=768 + acc")
timeit(lambda: torch.addmm(res, g768, ws[768]), "  lib: addmm")
timeit(lambda: rowgemm(x, ws[256], w_is_nk=False, lnb=(x, gam, mean, rstd), lnb_save=True), "LNbwd + dgrad N=256")
wt = {n: ws[n].t().contiguous() for n in ws}
w2048t = w2048.t().contiguous()
timeit(lambda: rowgemm(x, wt[768], w_is_nk=False, pos=pos, pos_cols=512), "KN qkv N=768 K=256 (+pos)")
timeit(lambda: rowgemm(x, wt[256], w_is_nk=False, res=res, ln=(gam, bet, 1e-5), ln_save=True), "KN out+res+LN K=256")
timeit(lambda: rowgemm(h, w2048t, w_is_nk=False, res=res, ln=(gam, bet, 1e-5), ln_save=True), "KN linear2+res+LN K=2048")
timeit(lambda: rowgemm(x, wt[2048], w_is_nk=False, relu=True), "KN linear1+relu N=2048")
timeit(lambda: rowgemm(x, wt[384], w_is_nk=False, pos=pos), "KN msda qproj N=384 (+pos)")
timeit(lambda: rowgemm(x, wt[256], w_is_nk=False, pos=pos), "KN q N=256 (+pos)")
timeit(lambda: torch.add(x, pos), "  one add kernel")
timeit(lambda: None if torch.addmm(res, x, wt[256]) is None else None, "  lib: addmm 256")
timeit(lambda: rowgemm(x, w2048, w_is_nk=False, mask=h, lnb=(x, gam, mean, rstd), lnb_save=True), "LNbwd + dgrad + dReLU N=2048")
timeit(lambda: rowgemm(h, ws[2048], w_is_nk=False, res=res), "dgrad K=2048 + acc")
timeit(lambda: torch.addmm(res, h, ws[2048]), "  lib: addmm")
