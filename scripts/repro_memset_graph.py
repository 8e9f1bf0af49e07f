"""Minimal repro (ROCm 7.2 / torch 2.10, gfx950): a hipMemsetAsync captured into a hipGraph is replayed out of order
with the kernels around it.  Graph: x := 5 (kernel); memset(x, 0); x += 1 (kernel); y := x.  In order, y == 1 after every
replay.  Exit code 0: in order; 1: fault reproduced (prints what y held).  Run as a child process (it touches the GPU)."""
import ctypes
import sys

import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x = torch.zeros(n, dtype=torch.int32, device="cuda")
y = torch.zeros_like(x)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        x.fill_(5)
        rc = hip.hipMemsetAsync(x.data_ptr(), 0, 4 * n, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        x.add_(1)
        y.copy_(x)
bad = []
for i in range(5):
    g.replay()
    torch.cuda.synchronize()
    vals = sorted(set(y.cpu().tolist()))
    if vals != [1]:
        bad.append((i, vals[:4]))
print("memset node in a hipGraph:", "OUT OF ORDER, y held %s (want [1])" % bad if bad else "in order")
sys.exit(1 if bad else 0)
