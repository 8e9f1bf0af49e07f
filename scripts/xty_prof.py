import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import dense
dev = torch.device("cuda"); B, N, d, a = 2, 22223, 256, 64
v = torch.randn(B, N, d, device=dev); P = torch.randn(B, a, N, device=dev); G = torch.randn(B, N, a, device=dev)
for _ in range(10):
    dense.xty(P, v, x_transposed=True); dense.xty(v, G); dense.xty(G, v)
    torch.bmm(P, v); torch.bmm(v.transpose(1, 2), G)
torch.cuda.synchronize()
