#!/usr/bin/env python3
"""Developer micro-benchmark of the two native entry points (decoder + encoder shapes).
Interleaved rounds in one process; prints median/min microseconds and algorithmic GB/s."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import (NORTH_STAR_SHAPES, encoder_loc, graphed, make_msda_inputs,  # noqa: E402
                   msda_algorithmic_bytes, timeit)
from ziragroundingdino_amd import _C  # noqa: E402


def main():
    dev = torch.device("cuda")
    rounds = int(os.environ.get("ROUNDS", "7"))
    cfgs = []
    B, M, D, P = 2, 8, 32, 4
    shapes = NORTH_STAR_SHAPES
    S = sum(h * w for h, w in shapes)
    v, sh, st, loc, attn, go = make_msda_inputs(B, 900, M, D, shapes, P, 0, dev)
    cfgs.append(("decoder_uniform", (v, sh, st, loc, attn, go), 900, 200))
    g = torch.Generator().manual_seed(1)
    centre = torch.rand(B, 900, 1, 1, 1, 2, generator=g) * 0.8 + 0.1
    cl = (centre + 0.05 * torch.randn(B, 900, M, 4, P, 2, generator=g)).to(dev)
    cfgs.append(("decoder_clustered", (v, sh, st, cl, attn, go), 900, 200))
    # committed capture of a training step's decoder inputs (tests/golden), and the hot / pinpoint patterns of the tests
    gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "inmodel_decoder_locations.npz")
    with np.load(gold) as z:
        il, ia = torch.from_numpy(z["loc"].astype(np.float32)).to(dev), torch.from_numpy(z["attn"].astype(np.float32)).to(dev)
    cfgs.append(("decoder_inmodel", (v, sh, st, il, ia, go), 900, 200))
    rng = np.random.default_rng(5)
    spots = rng.uniform(0.2, 0.8, (3, 2))
    ctr = spots[rng.integers(0, 3, (B, 900))][:, :, None, None, None, :]
    for nm, sd in (("decoder_hot", 0.01), ("decoder_pinpoint", 0.0)):
        hl = torch.from_numpy((ctr + sd * rng.standard_normal((B, 900, M, 4, P, 2))).astype(np.float32)).to(dev)
        cfgs.append((nm, (v, sh, st, hl, attn, go), 900, 200))
    ve, _, _, _, attne, goe = make_msda_inputs(B, S, M, D, shapes, P, 2, dev)
    cfgs.append(("encoder", (ve, sh, st, encoder_loc(B, M, shapes, P, 3, dev), attne, goe), S, 20))
    if not os.environ.get("CASES"):  # measured streaming ceiling beside the 8 TB/s spec figure
        src = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()   # 1 GiB
        dst = torch.empty_like(src)
        for _ in range(3):
            dst.copy_(src)
        t = min(timeit(lambda: dst.copy_(src), 10) for _ in range(3))
        print("device-to-device copy of 1 GiB: %.1f us -> %.0f GB/s read+write" % (t, 2 * src.numel() * 4 / t / 1e3), flush=True)
        del src, dst
    if os.environ.get("ZIRA_INPUTS"):  # MSDA inputs captured from a model step (scripts/inmodel_msda.py)
        for key, t in torch.load(os.environ["ZIRA_INPUTS"]).items():
            t = [x.to(dev) for x in t]
            cfgs.append(("inmodel_" + key, tuple(t), t[3].shape[1], 20 if key == "enc" else 200))
    only = os.environ.get("CASES")
    for name, (v, sh, st, loc, attn, go), Q, iters in cfgs:
        if only and not any(o in name for o in only.split(",")):
            continue
        fb, bb = msda_algorithmic_bytes(B, S, M, D, 4, Q, P)
        fwd = lambda: _C.ms_deform_attn_forward(v, sh, st, loc, attn, 64)
        bwd = lambda: _C.ms_deform_attn_backward(v, sh, st, loc, attn, go, 64)
        for _ in range(3):
            fwd(); bwd()
        per = 20
        gf, gb = graphed(fwd, per), graphed(bwd, per)
        tf, tb = [], []
        for _ in range(rounds):
            tf.append(timeit(gf, max(1, iters // per)) / per)
            tb.append(timeit(gb, max(1, iters // per)) / per)
        print("%-18s fwd med %8.2f us (min %8.2f) %7.0f GB/s | bwd med %8.2f us (min %8.2f) %7.0f GB/s"
              % (name, np.median(tf), min(tf), fb / np.median(tf) / 1e3,
                 np.median(tb), min(tb), bb / np.median(tb) / 1e3), flush=True)
        plan = _C.ms_deform_attn_plan(v, sh, st, loc, attn, 64)
        if plan is not None:   # sparse calls: the plan made right behind the forward gather, the backward from the plan
            fwdp = lambda: _C.ms_deform_attn_forward_plan(v, sh, st, loc, attn, 64)
            bwdp = lambda: _C.ms_deform_attn_backward(v, sh, st, loc, attn, go, 64, plan=plan)
            planonly = lambda: _C.ms_deform_attn_plan(v, sh, st, loc, attn, 64)
            for _ in range(3):
                fwdp(); bwdp()
            gfp, gbp, gpo = graphed(fwdp, per), graphed(bwdp, per), graphed(planonly, per)
            tf, tb, tp = [], [], []
            for _ in range(rounds):
                tf.append(timeit(gfp, max(1, iters // per)) / per)
                tb.append(timeit(gbp, max(1, iters // per)) / per)
                tp.append(timeit(gpo, max(1, iters // per)) / per)
            print("%-18s fwd+plan med %6.2f us (min %6.2f) | planned bwd med %6.2f us (min %6.2f) | pair %6.2f us = %.3f of 8 TB/s | plan alone %6.2f us"
                  % (name, np.median(tf), min(tf), np.median(tb), min(tb), np.median(tf) + np.median(tb),
                     (fb + bb) / (np.median(tf) + np.median(tb)) / 1e3 / 8000.0, np.median(tp)), flush=True)

if __name__ == "__main__":
    main()
