"""Error statistics of the split-bf16 GEMM beside the library's fp32 GEMM, against an fp64 product (developer script).
    python scripts/gemm_bf16x3_accuracy.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import gemm_bf16x3 as g3
torch.manual_seed(2)
dev = "cuda"
for M, N, K, kind in ((8192, 2048, 256, "signed"), (8192, 256, 2048, "signed"), (8192, 256, 2048, "relu"), (8192, 256, 2048, "positive both"),
                      (44446, 256, 2048, "relu"), (8192, 256, 8192, "signed")):
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    if kind == "relu":
        a = a.relu_()
    if kind == "positive both":
        a, w = a.abs_(), w.abs_()
    ref = a.double() @ w.double().t()
    absref = a.double().abs() @ w.double().abs().t()
    lib = (a @ w.t()).double()
    ours = g3.gemm(a, g3.split_planes(w, False), g3.EPI_ADD, aux=torch.zeros(M, N, device=dev)).double()
    scale = float(ref.abs().max())
    def st(x):
        e = x - ref
        return "max %.2e rms %.2e bias %+.2e | rel to sum|ab|: max %.2e rms %.2e" % (
            float(e.abs().max()) / scale, float(e.pow(2).mean().sqrt()) / scale, float(e.mean()) / scale,
            float((e.abs() / absref).max()), float((e / absref).pow(2).mean().sqrt()))
    print("M %5d N %4d K %4d %-14s scale %.3g\n    ours %s\n    lib  %s" % (M, N, K, kind, scale, st(ours), st(lib)), flush=True)
