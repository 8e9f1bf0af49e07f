import torch, time
dev = torch.device("cuda")
x = torch.randn(44446, 256, device=dev); w = torch.randn(2048, 256, device=dev) * 0.05; b = torch.randn(2048, device=dev)
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for lib in ("default", "hipblaslt", "rocblas"):
    if lib != "default":
        try:
            torch.backends.cuda.preferred_blas_library(lib)
        except Exception as e:
            print(lib, "not selectable:", e); continue
    print("[%s] addmm            %.1f us" % (lib, t(lambda: torch.addmm(b, x, w.t()))))
    print("[%s] addmm + relu     %.1f us" % (lib, t(lambda: torch.relu(torch.addmm(b, x, w.t())))))
    print("[%s] _addmm_activation %.1f us" % (lib, t(lambda: torch._addmm_activation(b, x, w.t()))))
    print("[%s] linear+relu_     %.1f us" % (lib, t(lambda: torch.nn.functional.linear(x, w, b).relu_())))
