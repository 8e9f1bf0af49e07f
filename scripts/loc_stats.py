"""Developer: distribution of the sampling locations of captured model inputs (ZIRA_INPUTS)."""
import os, sys, torch
d = torch.load(os.environ.get("ZIRA_INPUTS", "/tmp/inmodel.pt"))
for key in ("enc", "dec"):
    v, sh, st, loc, attn, go = d[key]
    B, Q, M, L, P, _ = loc.shape
    print(key, "Q", Q, "attn min/max %.4f %.4f" % (attn.min().item(), attn.max().item()))
    for l in range(L):
        H, W = [int(x) for x in sh[l]]
        x = loc[:, :, :, l, :, 0] * W - 0.5
        y = loc[:, :, :, l, :, 1] * H - 0.5
        valid = (x > -1) & (y > -1) & (x < W) & (y < H)
        fx, fy = x - x.floor(), y - y.floor()
        cell = ((y.floor().clamp(-1, H - 1) + 1) * (W + 1) + x.floor().clamp(-1, W - 1) + 1).long()
        per_head = []
        for m in range(M):
            c = torch.bincount(cell[0, :, m][valid[0, :, m]].flatten(), minlength=(H + 1) * (W + 1)).float()
            per_head.append((c.max().item(), c.mean().item(), (c > 0).float().mean().item()))
        print("  level %d (%dx%d): valid %.1f%%; frac x in {0}: %.1f%%, distinct frac values ~%d; per head cells max/mean/hit: %s"
              % (l, H, W, 100 * valid.float().mean().item(), 100 * (fx.abs() < 1e-6).float().mean().item(),
                 len(torch.unique((fx[0, :200, 0] * 1000).round())),
                 " ".join("%d/%.1f/%.0f%%" % (a, b_, 100 * c_) for a, b_, c_ in per_head[:4])))
