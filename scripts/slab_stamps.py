"""Developer: per-phase stamps of the slab-scan backward (build with -DZIRA_SLAB_STAMPS=1)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import NORTH_STAR_SHAPES, make_msda_inputs
from ziragroundingdino_amd import _C, _lib
lib = _lib.load()
dev = torch.device("cuda")
shape = sys.argv[1] if len(sys.argv) > 1 else "decoder"
v, sh, st, loc, attn, go = make_msda_inputs(2, 900, 8, 32, NORTH_STAR_SHAPES, 4, 0, dev)
if shape == "clustered":
    g = torch.Generator().manual_seed(1)
    centre = torch.rand(2, 900, 1, 1, 1, 2, generator=g) * 0.8 + 0.1
    loc = (centre + 0.05 * torch.randn(2, 900, 8, 4, 4, 2, generator=g)).to(dev)
if shape.startswith("inmodel_"):
    v, sh, st, loc, attn, go = [t.to(dev) for t in torch.load(os.environ.get("ZIRA_INPUTS", "/tmp/inmodel.pt"))[shape[8:]]]
for _ in range(3):
    _C.ms_deform_attn_backward(v, sh, st, loc, attn, go, 64)
torch.cuda.synchronize()
N = 4096
buf = (ctypes.c_ulonglong * (8 * N))()
lib.zira_dev_read_slab_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.zira_dev_read_slab_stamps(buf, 8 * N) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(N, 8).astype(np.int64)
a = a[a[:, 6] > 0]
t0 = a[:, 0].min()
print("blocks stamped %d; span %.2f us" % (len(a), (a[:, 6].max() - t0) / 100.0))
names = ["pass A (hist)", "home dots", "prefix", "zero+pass B", "row sums", "fold"]
d = np.diff(a[:, :7], axis=1) / 100.0
lvl = a[:, 7] & 0xff
nh = a[:, 7] >> 32
for l in sorted(set(lvl)):
    m = lvl == l
    print("level %d: blocks %d, home samples mean %.0f max %d; start p50 %.1f; end p50 %.1f max %.1f"
          % (l, m.sum(), nh[m].mean(), nh[m].max(), np.median(a[m, 0] - t0) / 100.0, np.median(a[m, 6] - t0) / 100.0, (a[m, 6] - t0).max() / 100.0))
    for i, nme in enumerate(names):
        print("    %-14s mean %6.2f us  p50 %6.2f  max %6.2f" % (nme, d[m, i].mean(), np.median(d[m, i]), d[m, i].max()))
