"""Cold and warm time of the decoder MSDA call pair (forward + plan, planned backward) for ONE library build
(ZIRA_MSDA_LIB=build_ab/<name>.so): what bench.py reports as roofline.frac, without the training step around it.
    python scripts/cold_pair.py [inmodel|uniform ...]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import NORTH_STAR_SHAPES, graphed, make_msda_inputs, msda_algorithmic_bytes, msda_call_pair, msda_cold_cycle, timeit
from ziragroundingdino_amd import _C
dev = torch.device("cuda")
v, sh, st, loc, attn, go = make_msda_inputs(2, 900, 8, 32, NORTH_STAR_SHAPES, 4, 0, dev)
fb, bb = msda_algorithmic_bytes(2, v.shape[1], 8, 32, 4, 900, 4)
tag = os.path.basename(os.environ.get("ZIRA_MSDA_LIB", "shipped"))
for what in (sys.argv[1:] or ["inmodel", "uniform"]):
    l, a = loc, attn
    if what == "inmodel":
        with np.load(os.path.join(ROOT, "tests", "golden", "inmodel_decoder_locations.npz")) as z:
            l, a = torch.from_numpy(z["loc"].astype(np.float32)).to(dev), torch.from_numpy(z["attn"].astype(np.float32)).to(dev)
    cf, cb = [], []
    for _ in range(3):
        f, b = msda_cold_cycle(_C, v, sh, st, l, a, go, sets=8, reps=10)
        cf.append(f); cb.append(b)
    fwd, bwd = msda_call_pair(_C, v, sh, st, l, a, go)
    gf, gb = graphed(fwd, 10), graphed(bwd, 10)
    timeit(gf, 2); timeit(gb, 2)
    wf, wb = timeit(gf, 20) / 10, timeit(gb, 20) / 10
    f, b = float(np.median(cf)), float(np.median(cb))
    print("%-22s %-8s cold fwd %5.2f bwd %5.2f pair %5.2f us = %.3f | warm fwd %5.2f bwd %5.2f pair %5.2f = %.3f"
          % (tag, what, f, b, f + b, (fb + bb) / (f + b) / 1e3 / 8000.0, wf, wb, wf + wb, (fb + bb) / (wf + wb) / 1e3 / 8000.0), flush=True)
