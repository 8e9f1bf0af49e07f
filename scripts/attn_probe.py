import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import transformer as T
from bench import timeit, graphed
dev = torch.device("cuda")
mha = torch.nn.MultiheadAttention(256, 8, dropout=0.0).to(dev)
for p in mha.parameters(): p.requires_grad_(False)
for (L, S, name) in ((900, 900, "self"), (900, 32, "text")):
    q = torch.randn(L, 2, 256, device=dev, requires_grad=True)
    kv = torch.randn(S, 2, 256, device=dev, requires_grad=True)
    g = torch.randn(L, 2, 256, device=dev)
    def fwd():
        return T.lean_mha(mha, q, kv if S != L else q, kv if S != L else q)
    def fb():
        o = fwd()
        torch.autograd.grad(o, [q] + ([kv] if S != L else []), g)
    for _ in range(3): fb()
    with torch.no_grad():
        tf = timeit(graphed(lambda: fwd(), 5), 20) / 5
    tfb = timeit(graphed(fb, 5), 20) / 5
    print("%s attention L=%d S=%d: fwd %.1f us, fwd+bwd %.1f us (incl. the three projections and out_proj)" % (name, L, S, tf, tfb))
    # the core alone
    qh = torch.randn(2, 8, L, 32, device=dev, requires_grad=True); kh = torch.randn(2, 8, S, 32, device=dev, requires_grad=True); vh = torch.randn(2, 8, S, 32, device=dev, requires_grad=True)
    go = torch.randn(2, 8, L, 32, device=dev)
    def core(): return T._attention_small(qh, kh, vh, None)
    def core_fb():
        torch.autograd.grad(core(), [qh, kh, vh], go)
    for _ in range(3): core_fb()
    with torch.no_grad():
        tc = timeit(graphed(lambda: core(), 5), 20) / 5
    tcb = timeit(graphed(core_fb, 5), 20) / 5
    print("   core (scores, softmax, PV): fwd %.1f us, fwd+bwd %.1f us" % (tc, tcb))
