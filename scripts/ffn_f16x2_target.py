"""Times the fused frozen-FFN launches (csrc/ffn_f16x2.hip) on the encoder shape beside what they replace: the library's fp32
GEMMs (+ csrc/gemm_drelu.hip in the backward) and the split-bf16 products (csrc/gemm_bf16x3.hip).  HIP events, 20 timed calls."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import ffn_f16x2 as ff, gemm_bf16x3 as g3, _lib, tuned_gemm
if os.environ.get("TUNED", "1") == "1":
    tuned_gemm.enable()

M = int(sys.argv[1]) if len(sys.argv) > 1 else 44446
F = 2048
torch.manual_seed(0)
x = torch.randn(M, 256, device="cuda")
w1 = torch.randn(F, 256, device="cuda") * 0.06
b1 = torch.randn(F, device="cuda") * 0.1
w2 = torch.randn(256, F, device="cuda") * 0.03
b2 = torch.randn(256, device="cuda") * 0.1
gy = torch.randn(M, 256, device="cuda")


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


pk = ff.PackedFFN()
mask = ff.mask_like(x, F)
pf, pb = pk.get(w1, b1, w2, False), pk.get(w1, b1, w2, True)
out = torch.empty_like(x)
gs = gy.clone()
flop = 2 * 2.0 * M * 256 * F
t_f = timed(lambda: ff.run(x, pf, F, False, mask, q_bias=b2, out=out))
t_b = timed(lambda: ff.run(gy, pb, F, True, mask, aux=gs, out=gs))
print("f16x2 fused   forward %7.1f us (%5.1f TF/s fp32-equivalent)   backward %7.1f us (%5.1f TF/s)" % (t_f, flop / t_f * 1e-6, t_b, flop / t_b * 1e-6))
t_f = timed(lambda: ff.run(x, pf, F, False, mask, q_bias=b2, out=out, use_workspace=False))
t_b = timed(lambda: ff.run(gy, pb, F, True, mask, aux=gs, out=gs, use_workspace=False))
print("  whole blocks forward %7.1f us (%5.1f TF/s fp32-equivalent)   backward %7.1f us (%5.1f TF/s)" % (t_f, flop / t_f * 1e-6, t_b, flop / t_b * 1e-6))
for mm in (32768, 65536):
    xx, oo, mk = torch.randn(mm, 256, device="cuda"), torch.empty(mm, 256, device="cuda"), torch.empty(mm, F // 32, device="cuda", dtype=torch.int32)
    t = timed(lambda: ff.run(xx, pf, F, False, mk, q_bias=b2, out=oo))
    print("  M = %d (whole rounds) forward %7.1f us (%5.1f TF/s)" % (mm, t, 4.0 * mm * 256 * F / t * 1e-6))

h = torch.empty(M, F, device="cuda")


def lib_fwd():
    hh = torch._addmm_activation(b1, x, w1.t())
    return torch.addmm(b2, hh, w2.t())


hh = torch._addmm_activation(b1, x, w1.t())
g = torch.empty_like(hh)
lib = _lib.load()


def lib_bwd():
    lib.zira_gemm_drelu_f32(gy.data_ptr(), w2.data_ptr(), hh.data_ptr(), M, F, 256, g.data_ptr(), torch.cuda.current_stream().cuda_stream)
    gs.addmm_(g, w1)


t_lf, t_lb = timed(lib_fwd), timed(lib_bwd)
print("library fp32  forward %7.1f us (%5.1f TF/s)                   backward %7.1f us (%5.1f TF/s)" % (t_lf, flop / t_lf * 1e-6, t_lb, flop / t_lb * 1e-6))

s1, s2, s2t, s1t = g3.SplitWeight(False), g3.SplitWeight(False), g3.SplitWeight(True), g3.SplitWeight(True)
p1, p2, p2t, p1t = s1.planes(w1), s2.planes(w2), s2t.planes(w2), s1t.planes(w1)


def b3_fwd():
    a = g3.gemm(x, p1, g3.EPI_BIAS_RELU, bias=b1)
    return g3.gemm(a, p2, g3.EPI_BIAS, bias=b2)


def b3_bwd():
    g3.gemm(gy, p2t, g3.EPI_MASK, aux=hh, out=g)
    g3.gemm(g, p1t, g3.EPI_ADD, aux=gs, out=gs)


t_3f, t_3b = timed(b3_fwd), timed(b3_bwd)
print("bf16x3        forward %7.1f us (%5.1f TF/s)                   backward %7.1f us (%5.1f TF/s)" % (t_3f, flop / t_3f * 1e-6, t_3b, flop / t_3b * 1e-6))
