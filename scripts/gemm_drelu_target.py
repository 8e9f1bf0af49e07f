#!/usr/bin/env python3
"""Dev: zira_gemm_drelu_f32 at the encoder FFN shape beside the mm + threshold_backward it replaces, for
`scripts/kstats_py.sh gd scripts/gemm_drelu_target.py [iters]`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import _lib  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
lib = _lib.load()
M, N, K = 44446, 2048, 256
A = torch.randn(M, K, device="cuda")
B = torch.randn(K, N, device="cuda")
H = torch.randn(M, N, device="cuda").clamp_min(0)
C = torch.empty(M, N, device="cuda")
for _ in range(iters):
    lib.zira_gemm_drelu_f32(A.data_ptr(), B.data_ptr(), H.data_ptr(), M, N, K, C.data_ptr(), torch.cuda.current_stream().cuda_stream)
    g = torch.ops.aten.threshold_backward(A @ B, H, 0)
torch.cuda.synchronize()
