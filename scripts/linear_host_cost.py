#!/usr/bin/env python3
"""Host cost of one small linear through different call paths (us per call, GPU idle-bound shapes)."""
import os, sys, time, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import tuned_gemm
dev = "cuda"
x3 = torch.randn(900, 2, 256, device=dev); x2 = x3.reshape(-1, 256)
w = torch.randn(256, 256, device=dev); b = torch.randn(256, device=dev); wt = w.t().contiguous()
def bench(name, fn, n=3000):
    for _ in range(200): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("   %-34s host %6.1f us/call   (incl. drain %6.1f)" % (name, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6), flush=True)
for tun in (False, True):
    if tun: tuned_gemm.enable()
    print("TunableOp", "on (tuning off, committed file)" if tun else "off")
    bench("F.linear 3-D, bias", lambda: F.linear(x3, w, b))
    bench("F.linear 2-D, bias", lambda: F.linear(x2, w, b))
    bench("F.linear 2-D, no bias", lambda: F.linear(x2, w))
    bench("torch.addmm(b, x, w.t())", lambda: torch.addmm(b, x2, w.t()))
    bench("torch.mm(x, wt)", lambda: torch.mm(x2, wt))
    bench("torch.mm(x, wt).add_(b)", lambda: torch.mm(x2, wt).add_(b))
    bench("torch.addmm(b, x, wt)  [NN]", lambda: torch.addmm(b, x2, wt))
    bench("x2 + x2 (elementwise)", lambda: x2 + x2)
    xs = torch.randn(24, 256, device=dev)                      # text-sized: pure host cost
    bench("tiny F.linear 2-D, bias", lambda: F.linear(xs, w, b))
    bench("tiny torch.mm(x, wt)", lambda: torch.mm(xs, wt))
    bench("tiny torch.mm(x, wt).add_(b)", lambda: torch.mm(xs, wt).add_(b))
    bench("tiny torch.addmm(b, x, wt)", lambda: torch.addmm(b, xs, wt))
    w8 = torch.randn(2048, 256, device=dev); b8 = torch.randn(2048, device=dev); w8t = w8.t().contiguous()
    bench("ffn1 F.linear 1800x256->2048", lambda: F.linear(x2, w8, b8))
    bench("ffn1 mm(x, wt).add_(b)", lambda: torch.mm(x2, w8t).add_(b8))
    bench("ffn1 addmm(b, x, wt)", lambda: torch.addmm(b8, x2, w8t))
    h = torch.randn(1800, 2048, device=dev); w9 = torch.randn(256, 2048, device=dev); w9t = w9.t().contiguous()
    bench("ffn2 F.linear 1800x2048->256", lambda: F.linear(h, w9, b))
    bench("ffn2 mm(x, wt).add_(b)", lambda: torch.mm(h, w9t).add_(b))
    bench("ffn2 addmm(b, x, wt)", lambda: torch.addmm(b, h, w9t))
