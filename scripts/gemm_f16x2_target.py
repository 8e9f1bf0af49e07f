"""The two-plane f16 GEMM (csrc/gemm_f16x2.hip) beside the three-plane bfloat16 one (csrc/gemm_bf16x3.hip) and the library's
fp32 GEMM on the model's frozen products outside the FFN: microseconds and effective TFLOP/s.   python scripts/gemm_f16x2_target.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import gemm_bf16x3 as g3, tuned_gemm
if os.environ.get("TUNED", "1") == "1":
    tuned_gemm.enable()
torch.manual_seed(0)


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


for what, M, N, K in (("value / output projection", 44446, 256, 256), ("query projection", 44446, 384, 256), ("query input grad", 44446, 256, 384),
                      ("swin s1 fc1", 134400, 384, 96), ("swin s1 fc2", 134400, 128, 384), ("swin s2 qkv", 33600, 640, 192),
                      ("swin s2 fc1", 33600, 768, 192), ("swin s2 fc2", 33600, 256, 768), ("swin s3 fc1", 8400, 1536, 384),
                      ("swin s3 fc2", 8400, 384, 1536), ("FFN linear1", 44446, 2048, 256), ("FFN linear2", 44446, 256, 2048)):
    a = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda") * 0.05
    b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    p3, p2 = g3.split_planes(w, False), g3.split_planes_f16x2(w, False)
    t_lib = timed(lambda: torch.addmm(b, a, w.t(), out=out))
    t3 = timed(lambda: g3.gemm(a, p3, g3.EPI_BIAS, bias=b, out=out))
    t2 = timed(lambda: g3.gemm_f16x2(a, p2, N, g3.EPI_BIAS, bias=b, out=out))
    fl = 2.0 * M * N * K
    panel = ""
    if g3.panel_supported(N, K):
        pf = g3.split_frags_f16x2(w, False)
        tp = timed(lambda: g3.gemm_f16x2_panel(a, pf, N, g3.EPI_BIAS, bias=b, out=out))
        panel = "  panel %6.1f (%5.1f; %.2f of the bytes at 4.95 TB/s)" % (tp, fl / tp * 1e-6, (4.0 * M * (N + K)) / 4.95e12 / (tp * 1e-6))
    print("%-26s M=%6d N=%4d K=%4d  library %6.1f us (%5.1f TF/s)  bf16x3 %6.1f (%5.1f)  f16x2 %6.1f (%5.1f)%s" % (
        what, M, N, K, t_lib, fl / t_lib * 1e-6, t3, fl / t3 * 1e-6, t2, fl / t2 * 1e-6, panel))
