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
full = len(sys.argv) > 2 and sys.argv[2] == "full"       # the bench configuration instead of the small test model
do_poison = not (len(sys.argv) > 3 and sys.argv[3] == "clean")
if full:
    from ziragroundingdino_amd.config import zira_swint_config
    from ziragroundingdino_amd.groundingdino import build_model
    torch.manual_seed(0)
    model = build_model(zira_swint_config(device="cuda")).to("cuda").train()
    batches = [synthetic_batch(2, 800, 1333, seed=s, device="cuda") for s in range(4)]
else:
    model = small_model().train()
    batches = [synthetic_batch(2, 224, 320, n_categories=4, boxes_per_image=3, seed=s, device="cuda") for s in range(4)]
tr = ZiraTrainer(model)
for i in range(6 if full else 12):
    if do_poison:
        poison()
    data = batches[i % 4]
    nxt = batches[(i + 1) % 4] if mode == "prefetch" else None
    out = tr.run_step(data, next_data=nxt) if nxt is not None else tr.run_step(data)
    torch.cuda.synchronize()
    bad = [k for k, v in out.items() if not torch.isfinite(v)]
    inf = lsap.infeasible("cuda", reset=True)
    print("step %d: total %.5f%s%s" % (i, float(sum(out.values())), "  NON-FINITE " + ",".join(bad) if bad else "", "  LSAP-INFEASIBLE" if inf else ""), flush=True)
print("done")
