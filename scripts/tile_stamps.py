"""Per-block phase times of msda_bwd_tile_accum (library built with -DZIRA_DEV_STAMPS=1, ZIRA_MSDA_LIB=...):
   python scripts/tile_stamps.py [uniform|inmodel]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import NORTH_STAR_SHAPES, make_msda_inputs
from ziragroundingdino_amd import _C, _lib
lib = _lib.load()
dev = torch.device("cuda")
v, sh, st, loc, attn, go = make_msda_inputs(2, 900, 8, 32, NORTH_STAR_SHAPES, 4, 0, dev)
if len(sys.argv) > 1 and sys.argv[1] == "inmodel":
    with np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "inmodel_decoder_locations.npz")) as z:
        loc, attn = torch.from_numpy(z["loc"].astype(np.float32)).to(dev), torch.from_numpy(z["attn"].astype(np.float32)).to(dev)
for _ in range(5):
    _C.ms_deform_attn_backward(v, sh, st, loc, attn, go, 64)
torch.cuda.synchronize()
n = 16 * 2048
buf = (ctypes.c_ulonglong * n)()
lib.zira_dev_read_tile_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.zira_dev_read_tile_stamps(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 16).astype(np.int64)
a = a[a[:, 8] > 0]
names = ["prologue (items, first loads, clear)", "steps", "barrier + next item's first loads", "flush + barrier"]
print("blocks with items:", len(a), " items per block mean %.2f max %d" % (a[:, 8].mean(), a[:, 8].max()))
tot = a[:, :4].sum(1) / 100.0
print("block busy time (us): mean %.2f  p50 %.2f  max %.2f" % (tot.mean(), np.median(tot), tot.max()))
for i, nme in enumerate(names):
    x = a[:, i] / 100.0
    print("  %-34s mean %6.2f us  max %6.2f   per item %5.2f" % (nme, x.mean(), x.max(), (a[:, i] / np.maximum(a[:, 8], 1)).mean() / 100.0))

order = np.argsort(-tot)
print("slowest / fastest blocks: busy us (prologue, steps, barrier, flush) items")
for i in list(order[:6]) + list(order[-3:]):
    print("   %6.2f  (%5.2f %5.2f %5.2f %5.2f)  %d" % (tot[i], a[i, 0] / 100.0, a[i, 1] / 100.0, a[i, 2] / 100.0, a[i, 3] / 100.0, a[i, 8]))
print("busy-time percentiles (us):", " ".join("%d%%=%.1f" % (q, np.percentile(tot, q)) for q in (10, 25, 50, 75, 90, 99)))
lib.zira_dev_read_plan_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.zira_dev_read_plan_stamps(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(2048, 16).astype(np.int64)
a = a[a[:, 8] > 0]
print("plan blocks:", len(a))
for i, nme in enumerate(["levels + clear", "pass 1 (loc, cells, ranks)", "scan + classes", "pass 2 (records)", "counts", "items"]):
    x = a[:, i] / 100.0
    print("  %-34s mean %6.2f us  max %6.2f" % (nme, x.mean(), x.max()))
