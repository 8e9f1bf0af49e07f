#!/usr/bin/env python3
"""Dev: per-tile entry counts of the sparse backward's tiling for given sampling locations (in-model capture or synthetic)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import NORTH_STAR_SHAPES, make_msda_inputs, clustered_loc

def counts(loc, shapes, T):
    B, Q, M, L, P, _ = loc.shape
    out = []
    for l, (H, W) in enumerate(shapes):
        x = loc[:, :, :, l, :, 0] * W - 0.5
        y = loc[:, :, :, l, :, 1] * H - 0.5
        x0, y0 = torch.floor(x).long(), torch.floor(y).long()
        span = (H * W + T - 1) // T
        cnt = torch.zeros(B, M, T, dtype=torch.long)
        valid = (x > -1) & (y > -1) & (x < W) & (y < H)
        for dy in (0, 1):
            for dx in (0, 1):
                xx, yy = x0 + dx, y0 + dy
                ok = valid & (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
                t = ((yy * W + xx).clamp(0, H * W - 1) // span)
                bm = (torch.arange(B)[:, None, None, None] * M + torch.arange(M)[None, None, :, None]).expand_as(t)
                idx = (bm * T + t)[ok]
                cnt.view(-1).index_add_(0, idx, torch.ones_like(idx))
        out.append(cnt)
    return torch.stack(out, 2)      # [B, M, L, T]

def report(name, loc, shapes, T=80):
    c = counts(loc.cpu().float(), shapes, T).view(-1)
    n = c.numel()
    print("%-12s tiles %d mean %.0f p50 %d p90 %d p99 %d max %d | >512: %d tiles, >384: %d, >256: %d | extra slices @512: %d @256(h384): %d" % (
        name, n, c.float().mean(), c.median(), c.kthvalue(int(n * .9)).values, c.kthvalue(int(n * .99)).values, c.max(),
        (c > 512).sum(), (c > 384).sum(), (c > 256).sum(),
        (((c + 511) // 512 - 1) * (c > 512)).sum(), (((c + 255) // 256 - 1) * (c > 384)).sum()))
    for G, order in ((8, "strided"), (8, "adjacent")):
        cc = c.view(8, -1)                                 # 8 XCD chunks of consecutive virtual tiles
        per = cc.shape[1]
        ng = (per + G - 1) // G
        pad = torch.zeros(8, ng * G, dtype=c.dtype); pad[:, :per] = cc
        grp = pad.view(8, G, ng).transpose(1, 2) if order == "strided" else pad.view(8, ng, G)
        extra = (((grp + 511) // 512 - 1) * (grp > 512)).sum(-1).view(-1)
        print("     groups of %d (%s): %d groups, extra slices per group mean %.2f p90 %d p99 %d max %d" % (
            G, order, extra.numel(), extra.float().mean(), extra.kthvalue(int(extra.numel() * .9)).values,
            extra.kthvalue(int(extra.numel() * .99)).values, extra.max()))

if __name__ == "__main__":
    B, M, D, P, Q = 2, 8, 32, 4, 900
    v, sh, st, loc, attn, go = make_msda_inputs(B, Q, M, D, NORTH_STAR_SHAPES, P, 0, "cpu")
    report("uniform", loc, NORTH_STAR_SHAPES)
    report("clustered", clustered_loc(B, Q, M, 4, P, 1, "cpu"), NORTH_STAR_SHAPES)
    if len(sys.argv) > 1:
        for k, t in torch.load(sys.argv[1]).items():
            if k.startswith("dec"):
                report("inmodel_" + k, t[3], NORTH_STAR_SHAPES)
                break
