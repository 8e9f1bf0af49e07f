"""A few launches of the fused FFN kernels at the encoder shape (for scripts/pmc_py.sh / rocprofv3 passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import ffn_f16x2 as ff
torch.manual_seed(0)
M, F = int(sys.argv[1]) if len(sys.argv) > 1 else 44446, 2048
w1 = torch.randn(F, 256, device="cuda") * 0.06; b1 = torch.randn(F, device="cuda") * 0.1
w2 = torch.randn(256, F, device="cuda") * 0.03; b2 = torch.randn(256, device="cuda") * 0.1
pk = ff.PackedFFN(); pf, pb = pk.get(w1, b1, w2, False), pk.get(w1, b1, w2, True)
x = torch.randn(M, 256, device="cuda"); o = torch.empty_like(x); mk = ff.mask_like(x, F); g = torch.randn(M, 256, device="cuda")
for _ in range(4):
    ff.run(x, pf, F, False, mk, q_bias=b2, out=o)
    ff.run(g, pb, F, True, mk, aux=g, out=g)
torch.cuda.synchronize()
