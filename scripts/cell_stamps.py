"""Developer: per-phase wall-clock stamps of the walk kernel (build with -DZIRA_CELL_STAMPS=1).
usage: ZIRA_MSDA_LIB=build_ab/stamps.so python scripts/cell_stamps.py {decoder|clustered|encoder}"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench import NORTH_STAR_SHAPES, make_msda_inputs
from ziragroundingdino_amd import _C, _lib
lib = _lib.load()
dev = torch.device("cuda")
shape = sys.argv[1] if len(sys.argv) > 1 else "decoder"
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
if shape.startswith("inmodel_"):  # captured from a model step by scripts/inmodel_msda.py (ZIRA_SAVE_INPUTS)
    v, sh, st, loc, attn, go = [t.to(dev) for t in torch.load(os.environ.get("ZIRA_INPUTS", "/tmp/inmodel.pt"))[shape[8:]]]
for _ in range(3):
    _C.ms_deform_attn_backward(v, sh, st, loc, attn, go, 64)
torch.cuda.synchronize()
assert lib.zira_dev_clear_cell_stamps() == 0
_C.ms_deform_attn_backward(v, sh, st, loc, attn, go, 64)
torch.cuda.synchronize()
N = 16384
buf = (ctypes.c_ulonglong * (16 * N))()
lib.zira_dev_read_cell_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.zira_dev_read_cell_stamps(buf, 16 * N) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(N, 16).astype(np.int64)
a = a[a[:, 7] > 0]
t0 = a[:, 0].min()
print("work items stamped: %d; kernel span %.2f us" % (len(a), (a[:, 7].max() - t0) / 100.0))
names = ["block start->item", "runs/prefix", "sweeps+scan", "walk prologue", "walk loop", "final advances", "drain+rest"]
d = np.diff(a[:, :8], axis=1) / 100.0
lvl = (a[:, 8] >> 16) & 0xff
K = (a[:, 8] >> 8) & 0xff
twl = a[:, 8] & 0xff
n = a[:, 8] >> 32
nvis = a[:, 9] >> 32
maxlen = a[:, 10]
for l in sorted(set(lvl)):
    m = lvl == l
    print("level %d: items %5d  K %d  tw %d | records/tile mean %.0f max %d | visits(pass 0) mean %.0f max %d | maxlen mean %.0f max %d"
          % (l, m.sum(), K[m][0], 1 << twl[m][0], n[m].mean(), n[m].max(), nvis[m].mean(), nvis[m].max(), maxlen[m].mean(), maxlen[m].max()))
    for i, nme in enumerate(names):
        print("    %-18s mean %8.2f us  p50 %8.2f  max %8.2f" % (nme, d[m, i].mean(), np.median(d[m, i]), d[m, i].max()))
    life = (a[m, 7] - a[m, 1]) / 100.0
    print("    item lifetime      mean %8.2f us  p50 %8.2f  max %8.2f ; end time p50 %.1f max %.1f us"
          % (life.mean(), np.median(life), life.max(), np.median(a[m, 7] - t0) / 100.0, (a[m, 7] - t0).max() / 100.0))
print("item start (us after kernel start): p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile((a[:, 1] - t0) / 100.0, [50, 90, 100])))
