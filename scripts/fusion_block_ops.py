import os, sys, torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, "/root/repo")
from ziragroundingdino_amd import transformer
dev = torch.device("cuda")
torch.manual_seed(0)
blk = transformer.BiAttentionBlock(v_dim=256, l_dim=256, embed_dim=1024, num_heads=4, dropout=0.0, drop_path=0.0).to(dev).train()
for p in blk.parameters():
    p.requires_grad_(False)
v = torch.randn(2, 22223, 256, device=dev, requires_grad=True)
l = torch.randn(2, 32, 256, device=dev, requires_grad=True)
mask_l = torch.zeros(2, 32, dtype=torch.bool, device=dev)
gv, gl = torch.randn_like(v), torch.randn_like(l)
def step():
    ov, ol = blk(v, l, attention_mask_v=None, attention_mask_l=mask_l)
    torch.autograd.grad([ov, ol], [v, l], [gv, gl])
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = [r for r in prof.key_averages(group_by_input_shape=True) if r.self_device_time_total > 3]
rows.sort(key=lambda r: -r.self_device_time_total)
for r in rows[:40]:
    print("%8.1f us x%-3d %-40s %s" % (r.self_device_time_total, r.count, r.key[:40], str(r.input_shapes)[:110]))
