#!/usr/bin/env python3
"""Tall-reduction GEMMs of the re-bracketed BiAttention: rocBLAS one-shot vs manual split-K via bmm."""
import torch, time
dev = torch.device("cuda")
B, N, d, a = 2, 22223, 256, 64
v = torch.randn(B, N, d, device=dev)
P = torch.randn(B, a, N, device=dev)      # probs_l layout [B, H*T, N]
G = torch.randn(B, N, a, device=dev)      # grad of scores [B, N, H*T]

def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6

def chunked_PV(P, v, C=256):
    n0 = (N // C) * C
    Pm = P[:, :, :n0].reshape(B, a, n0 // C, C).permute(0, 2, 1, 3)          # [B, nc, a, C] (view)
    vm = v[:, :n0].reshape(B, n0 // C, C, d)                                  # [B, nc, C, d]
    out = torch.matmul(Pm, vm).sum(1)                                         # [B, a, d]
    if n0 < N:
        out = out + torch.bmm(P[:, :, n0:], v[:, n0:])
    return out

def chunked_vTG(v, G, C=256):
    n0 = (N // C) * C
    vm = v[:, :n0].reshape(B, n0 // C, C, d).transpose(2, 3)                  # [B, nc, d, C]
    Gm = G[:, :n0].reshape(B, n0 // C, C, a)
    out = torch.matmul(vm, Gm).sum(1)
    if n0 < N:
        out = out + torch.bmm(v[:, n0:].transpose(1, 2), G[:, n0:])
    return out

print("P@v  one-shot   %.1f us" % t(lambda: torch.bmm(P, v)))
for C in (128, 256, 512, 1024):
    print("P@v  chunk %4d %.1f us  (max err %.2e)" % (C, t(lambda: chunked_PV(P, v, C)), (chunked_PV(P, v, C) - torch.bmm(P, v)).abs().max().item()))
print("vT@G one-shot   %.1f us" % t(lambda: torch.bmm(v.transpose(1, 2), G)))
for C in (128, 256, 512, 1024):
    print("vT@G chunk %4d %.1f us  (max err %.2e)" % (C, t(lambda: chunked_vTG(v, G, C)), (chunked_vTG(v, G, C) - torch.bmm(v.transpose(1, 2), G)).abs().max().item()))
# the wide ones for reference
A = torch.randn(B, d, a, device=dev)
print("v@A  (N x 256 @ 256 x 64)   %.1f us" % t(lambda: torch.bmm(v, A)))
Z = torch.randn(B, a, d, device=dev); Pv = torch.randn(B, N, a, device=dev)
print("Pv@Z (N x 64 @ 64 x 256)    %.1f us" % t(lambda: torch.bmm(Pv, Z)))
print("G@AT (N x 64 @ 64 x 256)    %.1f us" % t(lambda: torch.bmm(G, A.transpose(1, 2))))
print("PT@U (N x 64 @ 64 x 256), P transposed view  %.1f us" % t(lambda: torch.bmm(P.transpose(1, 2), Z)))
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import dense
print("xty  P@v  (x_transposed)   %.1f us" % t(lambda: dense.xty(P, v, x_transposed=True)))
print("xty  vT@G                  %.1f us" % t(lambda: dense.xty(v, G)))
Pv2 = torch.randn(B, N, a, device=dev); go = torch.randn(B, N, d, device=dev)
print("xty  pvT@gout (64x256)     %.1f us" % t(lambda: dense.xty(Pv2, go)))
