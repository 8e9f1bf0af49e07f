#!/usr/bin/env python3
"""Tiny target for rocprofv3 --pmc runs: `prof_target.py {fwd|bwd|both} {decoder|clustered|encoder} [iters]`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import NORTH_STAR_SHAPES, make_msda_inputs, msda_call_pair  # noqa: E402
from ziragroundingdino_amd import _C  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "both"
shape = sys.argv[2] if len(sys.argv) > 2 else "decoder"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda")
B, M, D, P = 2, 8, 32, 4
S = sum(h * w for h, w in NORTH_STAR_SHAPES)
Q = S if shape == "encoder" else 900
v, sh, st, loc, attn, go = make_msda_inputs(B, Q, M, D, NORTH_STAR_SHAPES, P, 0, dev)
if shape == "encoder":
    from kbench import encoder_loc
    loc = encoder_loc(B, M, NORTH_STAR_SHAPES, P, 3, dev)
elif shape == "clustered":
    g = torch.Generator().manual_seed(1)
    centre = torch.rand(B, Q, 1, 1, 1, 2, generator=g) * 0.8 + 0.1
    loc = (centre + 0.05 * torch.randn(B, Q, M, 4, P, 2, generator=g)).to(dev)
if os.environ.get("ZIRA_INPUTS"):  # captured from a model step (scripts/inmodel_msda.py): shape = dec | enc
    v, sh, st, loc, attn, go = [t.to(dev) for t in torch.load(os.environ["ZIRA_INPUTS"])[shape]]
fwd, bwd = msda_call_pair(_C, v, sh, st, loc, attn, go)   # (as the autograd Function issues them: sparse calls plan in the forward)
for _ in range(iters):
    if which in ("fwd", "both"):
        fwd()
    if which in ("bwd", "both"):
        bwd()
torch.cuda.synchronize()
