import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import ffn_f16x2 as ff
torch.manual_seed(0)
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
F = 2048
w1 = torch.randn(F, 256, device="cuda") * 0.06; b1 = torch.randn(F, device="cuda") * 0.1
w2 = torch.randn(256, F, device="cuda") * 0.03; b2 = torch.randn(256, device="cuda") * 0.1
pk = ff.PackedFFN(); pf, pb = pk.get(w1,b1,w2,False), pk.get(w1,b1,w2,True)
for M in (256*128, 44446):
    x = torch.randn(M,256,device="cuda"); o = torch.empty_like(x); mk = ff.mask_like(x,F); g = torch.randn(M,256,device="cuda")
    t1 = timed(lambda: ff.run(x,pf,F,False,mk,q_bias=b2,out=o))
    t2 = timed(lambda: ff.run(g,pb,F,True,mk,aux=g,out=g))
    print("%s M=%6d fwd %6.1f bwd %6.1f us" % (os.environ.get("ZIRA_MSDA_LIB", "default"), M, t1, t2))
