"""Expose reads of uninitialised memory: the caching allocator's free blocks are filled with NaN bit patterns before every
step, so any torch.empty buffer that is read before it is written shows up as NaN / garbage."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_model_gpu import small_model
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch
from ziragroundingdino_amd import lsap

def poison():
    torch.cuda.synchronize()
    bufs = []
    for n in (1 << 8, 1 << 10, 1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20, 1 << 22, 1 << 24, 1 << 26):
        for _ in range(6):
            bufs.append(torch.full((n,), float("nan"), device="cuda"))
    torch.cuda.synchronize()
    del bufs

mode = sys.argv[1] if len(sys.argv) > 1 else "prefetch"
model = small_model().train()
tr = ZiraTrainer(model)
batches = [synthetic_batch(2, 224, 320, n_categories=4, boxes_per_image=3, seed=s, device="cuda") for s in range(4)]
for i in range(12):
    poison()
    data = batches[i % 4]
    nxt = batches[(i + 1) % 4] if mode == "prefetch" else None
    out = tr.run_step(data, next_data=nxt) if nxt is not None else tr.run_step(data)
    torch.cuda.synchronize()
    bad = [k for k, v in out.items() if not torch.isfinite(v)]
    inf = lsap.infeasible("cuda", reset=True)
    print("step %d: total %.5f%s%s" % (i, float(sum(out.values())), "  NON-FINITE " + ",".join(bad) if bad else "", "  LSAP-INFEASIBLE" if inf else ""), flush=True)
print("done")
