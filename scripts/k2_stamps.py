import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import NORTH_STAR_SHAPES, make_msda_inputs
from ziragroundingdino_amd import _C, _lib
lib = _lib.load()
dev = torch.device("cuda")
if len(sys.argv) > 1:
    v, sh, st, loc, attn, go = [t.to(dev) for t in torch.load(sys.argv[1])[sys.argv[2] if len(sys.argv) > 2 else "dec"]]
else:
    v, sh, st, loc, attn, go = make_msda_inputs(2, 900, 8, 32, NORTH_STAR_SHAPES, 4, 0, dev)
for _ in range(3):
    _C.ms_deform_attn_backward(v, sh, st, loc, attn, go, 64)
torch.cuda.synchronize()
n = 8 * 8192
buf = (ctypes.c_ulonglong * n)()
lib.zira_dev_read_k2_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.zira_dev_read_k2_stamps(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.int64)[:6000]  # owners (helpers: 6000+)
a = a[a[:, 0] > 0]
a = a[a[:, 7] > 0]
t0 = a[:, 0].min()
print("blocks with stamps:", len(a), " kernel span (us):", (a[:, 7].max() - t0) / 100.0)
names = ["start->descprefix", "->pass1", "->rowscan", "->pass2", "->zerofill", "->rowsums", "->barrier"]
d = np.diff(a, axis=1) / 100.0
for i, nme in enumerate(names):
    print("  %-18s mean %7.2f us  p50 %7.2f  max %7.2f" % (nme, d[:, i].mean(), np.median(d[:, i]), d[:, i].max()))
print("  block lifetime     mean %7.2f us  max %7.2f" % (((a[:, 7] - a[:, 0]) / 100.0).mean(), ((a[:, 7] - a[:, 0]) / 100.0).max()))
print("  block start spread (us): p50 %.2f  p90 %.2f  max %.2f" % tuple(np.percentile((a[:, 0] - t0) / 100.0, [50, 90, 100])))
# wave-per-tile K2: lifetime and phases by level (sparse plan: T tiles per level = row stamps / heads / L)
if loc.shape[1] != v.shape[1]:
    B, Q, M, L = loc.shape[0], loc.shape[1], loc.shape[2], loc.shape[3]
    full = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.int64)
    nt = None
    for T in range(1, 400):
        if B * M * L * T >= (full[:, 0] > 0).sum() and B * M * L * T <= 8192:
            nt = L * T
            break
    if nt:
        ids = np.nonzero((full[:, 0] > 0) & (full[:, 7] > 0))[0]
        lvl = (ids % nt) // (nt // L)
        life = (full[ids, 7] - full[ids, 0]) / 100.0
        dd = np.diff(full[ids], axis=1) / 100.0
        for l in range(L):
            m = lvl == l
            if m.any():
                print("  level %d: %4d waves, lifetime mean %6.2f p90 %6.2f max %6.2f | phases mean %s" % (
                    l, m.sum(), life[m].mean(), np.percentile(life[m], 90), life[m].max(),
                    " ".join("%5.2f" % x for x in dd[m].mean(0))))
# helper launch (stamp ids 6000 + wave): when do the slices start and how long do their phases take
full = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.int64)
own = full[:6000]; own = own[(own[:, 0] > 0) & (own[:, 7] > 0)]
hel = full[6000:]; hel = hel[(hel[:, 0] > 0) & (hel[:, 7] > 0)]
if len(hel) and len(own):
    t_own0, t_own1 = own[:, 0].min(), own[:, 7].max()
    busy = hel[hel[:, 6] > hel[:, 0]]
    print("owners: %.2f us span; helper waves stamped %d (with a slice: %d); first helper start %.2f us after the owners' end, helper span %.2f us"
          % ((t_own1 - t_own0) / 100.0, len(hel), len(busy), (hel[:, 0].min() - t_own1) / 100.0, (hel[:, 7].max() - hel[:, 0].min()) / 100.0))
    if len(busy):
        dd = np.diff(busy, axis=1) / 100.0
        print("  helper phases mean: " + " ".join("%5.2f" % x for x in dd.mean(0)) + " | max: " + " ".join("%5.2f" % x for x in dd.max(0)))
