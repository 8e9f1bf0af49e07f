#!/usr/bin/env python3
"""Entries per grad_value tile of a sparse MSDA call (the wave-per-tile K2 plan): how many tiles are
heavy, how many slices the helper launch gets.  usage: tile_hist.py <inputs.pt> [dec] [T]"""
import sys
import torch
v, sh, st, loc, attn, go = torch.load(sys.argv[1])[sys.argv[2] if len(sys.argv) > 2 else "dec"]
B, Q, M, L, P, _ = loc.shape
T = int(sys.argv[3]) if len(sys.argv) > 3 else (256 * 4 * 5) // (B * M * L)
tot_heavy = tot_slices = 0
for l in range(L):
    H, W = [int(x) for x in sh[l]]
    span = (H * W + T - 1) // T
    x = loc[:, :, :, l, :, 0].double() * W - 0.5
    y = loc[:, :, :, l, :, 1].double() * H - 0.5
    ok = (x > -1) & (y > -1) & (x < W) & (y < H)
    x0, y0 = x.floor().long(), y.floor().long()
    counts = torch.zeros(B * M * T, dtype=torch.long)
    head = (torch.arange(B)[:, None, None, None] * M + torch.arange(M)[None, None, :, None]).expand(B, Q, M, P)
    for dx in (0, 1):
        for dy in (0, 1):
            xx, yy = x0 + dx, y0 + dy
            m = ok & (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            tile = (yy * W + xx)[m] // span
            counts.index_add_(0, head[m] * T + tile, torch.ones_like(tile))
    heavy = counts > 512
    slices = ((counts[heavy] - 1) // 512).sum().item()
    tot_heavy += int(heavy.sum()); tot_slices += slices
    print("level %d (%dx%d, %d rows/tile): tiles %d, entries mean %.0f p50 %.0f p99 %.0f max %d, heavy %d, helper slices %d"
          % (l, H, W, span, counts.numel(), counts.float().mean(), counts.float().median(),
             counts.float().quantile(0.99), counts.max(), heavy.sum(), slices))
print("total: heavy tiles %d, helper slices %d (of %d tiles)" % (tot_heavy, tot_slices, B * M * T * L))
