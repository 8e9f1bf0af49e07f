"""Timeline of ONE msda_bwd_tile_accum launch by block (library built with scripts/build_variant.sh btimes -DZIRA_DEV_BTIMES=1,
selected with ZIRA_MSDA_LIB=build_ab/btimes.so): when the accumulate blocks finish their prologue / their last item, when the
gather blocks run and where.      python scripts/tile_timeline.py [uniform|inmodel] [cold|warm]
cold: eight independent operand sets (> 1 GB) are cycled through, the timeline is that of the last call (its operands and
its plan come from HBM, as inside a training step); warm: one operand set, back to back."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import NORTH_STAR_SHAPES, make_msda_inputs
from ziragroundingdino_amd import _C, _lib
lib = _lib.load()
dev = torch.device("cuda")
what = sys.argv[1] if len(sys.argv) > 1 else "inmodel"
mode = sys.argv[2] if len(sys.argv) > 2 else "cold"
v, sh, st, loc, attn, go = make_msda_inputs(2, 900, 8, 32, NORTH_STAR_SHAPES, 4, 0, dev)
if what == "inmodel":
    with np.load(os.path.join(ROOT, "tests", "golden", "inmodel_decoder_locations.npz")) as z:
        loc, attn = torch.from_numpy(z["loc"].astype(np.float32)).to(dev), torch.from_numpy(z["attn"].astype(np.float32)).to(dev)
sets = 8 if mode == "cold" else 1
copies = [tuple(t.clone() for t in (v, loc, attn, go)) for _ in range(sets)]
plans = [_C.ms_deform_attn_forward_plan(cv, sh, st, cl, ca, 64)[1] for cv, cl, ca, _ in copies]
keep = []
for rep in range(4):
    keep.clear()
    for (cv, cl, ca, cg), p in zip(copies, plans):
        keep.append(_C.ms_deform_attn_backward(cv, sh, st, cl, ca, cg, 64, plan=p))
torch.cuda.synchronize()
n = 8 * 8192
buf = (ctypes.c_ulonglong * n)()
lib.zira_dev_read_block_times.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.zira_dev_read_block_times(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.int64)
a = a[a[:, 2] > 0]
kind, xcc = a[:, 3] & 255, (a[:, 3] >> 8) & 15
t0 = a[:, 0].min()
us = lambda x: (x - t0) / 100.0
acc, gat = a[kind == 1], a[kind == 2]
print("%s / %s: %d accumulate blocks with items, %d gather blocks; last block ends at %.2f us" % (what, mode, len(acc), len(gat), us(a[:, 2].max())))
pct = lambda x: " ".join("%d%%=%.1f" % (q, np.percentile(x, q)) for q in (0, 10, 50, 90, 99, 100))
print("accumulate: start      ", pct(us(acc[:, 0])))
print("accumulate: prologue   ", pct((acc[:, 1] - acc[:, 0]) / 100.0), "(start -> first step)")
print("accumulate: end        ", pct(us(acc[:, 2])))
print("accumulate: busy       ", pct((acc[:, 2] - acc[:, 0]) / 100.0))
print("gather:     start      ", pct(us(gat[:, 0])))
print("gather:     end        ", pct(us(gat[:, 2])))
print("gather:     duration   ", pct((gat[:, 2] - gat[:, 0]) / 100.0))
if len(acc):   # what an accumulate block's time is made of: least squares over the blocks
    items, empty = (acc[:, 4] & 0xFFFFFFFF).astype(float), (acc[:, 4] >> 32).astype(float)
    steps, shares = (acc[:, 5] & 0xFFFFFFFF).astype(float), (acc[:, 5] >> 32).astype(float)
    busy = (acc[:, 2] - acc[:, 1]) / 100.0      # first step -> end
    X = np.stack([np.ones_like(items), items - empty, empty, steps, shares], 1)
    coef, *_ = np.linalg.lstsq(X, busy, rcond=None)
    print("busy after the prologue ~ %.2f + %.2f per non-empty item + %.2f per empty tile + %.2f per block step + %.2f per share  (rms residual %.2f us)"
          % (*coef, float(np.sqrt(np.mean((X @ coef - busy) ** 2)))))
    print("items per block:", pct(items), " steps per block:", pct(steps))
    order = np.argsort(-busy)
    for i in list(order[:5]) + list(order[-5:]):
        print("   busy %5.1f  items %3d (empty %3d)  steps %3d  shares %d" % (busy[i], items[i], empty[i], steps[i], shares[i]))
edges = np.arange(0, us(a[:, 2].max()) + 2.0, 2.0)
print("per 2 us: gather blocks started / accumulate blocks ended")
hs, _ = np.histogram(us(gat[:, 0]), edges)
he, _ = np.histogram(us(acc[:, 2]), edges)
for e, x, y in zip(edges, hs, he):
    print("  %5.1f  %5d  %5d" % (e, x, y))
print("by XCC: last accumulate end / last gather end:", " ".join("%d:%.1f/%.1f" % (x, us(acc[xcc[kind == 1] == x][:, 2].max()) if (xcc[kind == 1] == x).any() else 0,
      us(gat[xcc[kind == 2] == x][:, 2].max()) if (xcc[kind == 2] == x).any() else 0) for x in range(8)))
