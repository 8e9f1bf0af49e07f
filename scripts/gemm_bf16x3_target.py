"""The split-bf16 GEMM (csrc/gemm_bf16x3.hip) beside the library's fp32 GEMMs at the encoder FFN's four products
(M = 44446 image-token rows): microseconds and effective TFLOP/s (2 M N K / time).   python scripts/gemm_bf16x3_target.py [M]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import _lib, gemm_bf16x3 as g3
from ziragroundingdino_amd import tuned_gemm

if os.environ.get("TUNED", "1") == "1":
    tuned_gemm.enable()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 44446
dev = torch.device("cuda")
torch.manual_seed(0)
lib = _lib.load()


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


x, gs = torch.randn(M, 256, device=dev), torch.randn(M, 256, device=dev)
w1, b1 = torch.randn(2048, 256, device=dev) * 0.05, torch.randn(2048, device=dev)
w2, b2 = torch.randn(256, 2048, device=dev) * 0.02, torch.randn(256, device=dev)
h = torch._addmm_activation(b1, x, w1.t())
g = torch.randn(M, 2048, device=dev) * (h > 0)
p_w1, p_w2 = g3.split_planes(w1, False), g3.split_planes(w2, False)
p_w2t, p_w1t = g3.split_planes(w2, True), g3.split_planes(w1, True)
out_h, out_y, out_g, out_x = torch.empty_like(h), torch.empty(M, 256, device=dev), torch.empty_like(h), torch.empty(M, 256, device=dev)


def drelu():
    lib.zira_gemm_drelu_f32(gs.data_ptr(), w2.data_ptr(), h.data_ptr(), M, 2048, 256, out_g.data_ptr(), torch.cuda.current_stream().cuda_stream)


cases = [
    ("linear1 + bias + ReLU   [M,256]x[256,2048]", 2048, 256, lambda: torch._addmm_activation(b1, x, w1.t()),
     lambda: g3.gemm(x, p_w1, g3.EPI_BIAS_RELU, bias=b1, out=out_h)),
    ("linear2 + bias          [M,2048]x[2048,256]", 256, 2048, lambda: torch.addmm(b2, h, w2.t()),
     lambda: g3.gemm(h, p_w2, g3.EPI_BIAS, bias=b2, out=out_y)),
    ("grad @ W2 * (h > 0)     [M,256]x[256,2048]", 2048, 256, drelu,
     lambda: g3.gemm(gs, p_w2t, g3.EPI_MASK, aux=h, out=out_g)),
    ("gs += g @ W1            [M,2048]x[2048,256]", 256, 2048, lambda: out_x.addmm_(g, w1),
     lambda: g3.gemm(g, p_w1t, g3.EPI_ADD, aux=out_x, out=out_x)),
]
# the 256-wide projections of the deformable layer (value / output projections, the 384-wide query projection, their input gradients)
wv, bv = torch.randn(256, 256, device=dev) * 0.06, torch.randn(256, device=dev)
wq, bq = torch.randn(384, 256, device=dev) * 0.06, torch.randn(384, device=dev)
gproj = torch.randn(M, 384, device=dev)
p_wv, p_wvt, p_wq, p_wqt = g3.split_planes(wv, False), g3.split_planes(wv, True), g3.split_planes(wq, False), g3.split_planes(wq, True)
out_v, out_q, acc = torch.empty(M, 256, device=dev), torch.empty(M, 384, device=dev), torch.randn(M, 256, device=dev)
cases += [
    ("value_proj + bias       [M,256]x[256,256]", 256, 256, lambda: torch.addmm(bv, x, wv.t()),
     lambda: g3.gemm(x, p_wv, g3.EPI_BIAS, bias=bv, out=out_v)),
    ("query proj + bias       [M,256]x[256,384]", 384, 256, lambda: torch.addmm(bq, x, wq.t()),
     lambda: g3.gemm(x, p_wq, g3.EPI_BIAS, bias=bq, out=out_q)),
    ("gx += gv @ Wv           [M,256]x[256,256]", 256, 256, lambda: acc.addmm_(gs, wv),
     lambda: g3.gemm(gs, p_wvt, g3.EPI_ADD, aux=acc, out=acc)),
    ("gx += gproj @ Wq        [M,384]x[384,256]", 256, 384, lambda: acc.addmm_(gproj, wq),
     lambda: g3.gemm(gproj, p_wqt, g3.EPI_ADD, aux=acc, out=acc)),
]
tot = [0.0, 0.0]
for name, N, K, ref, ours in cases:
    tr, to = timeit(ref), timeit(ours)
    tot[0] += tr; tot[1] += to
    fl = 2.0 * M * N * K
    print("%-46s library %7.1f us (%5.1f TF/s)   bf16x3 %7.1f us (%5.1f TF/s)   x%.2f" % (name, tr, fl / tr / 1e6, to, fl / to / 1e6, tr / to), flush=True)
print("all products: library %.1f us, bf16x3 %.1f us" % tuple(tot))
