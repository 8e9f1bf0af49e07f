#!/usr/bin/env python3
"""Dev: every large GemmAndBias / Gemm entry of the committed TunableOp file, timed in place with TunableOp on and
off (cold operands: a large fill between calls) -- TunableOp's own timing has picked kernels that are slower in the
step (scripts/tall_linear.py).  Prints the entries where the tuned choice loses by more than 8 %."""
import os, re, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import tuned_gemm
from bench import timeit
dev = torch.device("cuda")
spoil = torch.empty(1 << 26, device=dev)
entries = []
for l in open(tuned_gemm.DEFAULT_FILE):
    m = re.match(r"(GemmAndBiasTunableOp|GemmTunableOp)_float_(TN|NN),(tn|nn)_(\d+)_(\d+)_(\d+)_ld_(\d+)_(\d+)_(\d+),", l)
    if m and int(m.group(5)) >= 4096:
        entries.append((m.group(1), m.group(2), int(m.group(4)), int(m.group(5)), int(m.group(6))))
def run(kind, lay, n, m, k):
    # column-major naming: C[n x m] ; row-major: out [m, n] = x [m, k] @ W^T ...
    x = torch.randn(m, k, device=dev)
    if lay == "TN":
        W = torch.randn(n, k, device=dev) * 0.05
        b = torch.randn(n, device=dev)
        fn = (lambda: F.linear(x, W, b)) if kind == "GemmAndBiasTunableOp" else (lambda: F.linear(x, W))
    else:
        Wt = torch.randn(k, n, device=dev) * 0.05
        fn = lambda: x @ Wt
    def cold():
        spoil.zero_(); fn()
    for _ in range(3): fn()
    tz = timeit(lambda: spoil.zero_(), 5)
    return min(timeit(cold, 5) for _ in range(3)) - tz
res = {}
for tuned in (False, True):
    (tuned_gemm.enable() if tuned else tuned_gemm.disable())
    for e in entries:
        res.setdefault(e, []).append(run(*e))
bad = 0
for e, (off, on) in sorted(res.items(), key=lambda kv: -kv[1][1] / kv[1][0]):
    flag = "  <-- tuned choice slower" if on > 1.08 * off else ""
    bad += bool(flag)
    print("%-22s %s n=%5d m=%6d k=%5d   default %7.1f us  tuned %7.1f us%s" % (*e, off, on, flag))
print("%d entries checked, %d where the tuned kernel loses" % (len(res), bad))
