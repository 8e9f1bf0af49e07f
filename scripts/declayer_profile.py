#!/usr/bin/env python3
"""Device time of one decoder layer forward+backward at the bench shape, by kernel (NATIVE=0: the module composition)."""
import os, sys, torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import transformer
from bench import NORTH_STAR_SHAPES
dev = torch.device("cuda"); torch.manual_seed(0)
layer = transformer.DeformableTransformerDecoderLayer(256, 2048, 0.0, "relu", 4, 8, 4, use_text_cross_attention=True).to(dev).train()
for p in layer.parameters(): p.requires_grad_(False)
S = sum(h * w for h, w in NORTH_STAR_SHAPES); Q, B, T = 900, 2, 16
tgt = torch.randn(Q, B, 256, device=dev, requires_grad=True)
qpos = torch.randn(Q, B, 256, device=dev)
memory_bf = torch.randn(B, S, 256, device=dev, requires_grad=True)   # (the encoder's output is batch-first)
memory = memory_bf.transpose(0, 1)
text = torch.randn(B, T, 256, device=dev, requires_grad=True)
tmask = torch.zeros(B, T, dtype=torch.bool, device=dev)
shapes = torch.tensor(NORTH_STAR_SHAPES, device=dev)
start = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
ref = torch.rand(Q, B, 4, 4, device=dev) * 0.5 + 0.25
g = torch.randn(Q, B, 256, device=dev)
NATIVE = os.environ.get("NATIVE", "1") == "1"
layer.native_layer = NATIVE
def step():
    # (the decoder projects the memory for all its layers at once: the layer's share is this GEMM and its gradient)
    mv = layer.cross_attn.value_proj(memory.transpose(0, 1))
    out = layer(memory_value=mv, tgt=tgt, tgt_query_pos=qpos, tgt_query_sine_embed=None, tgt_key_padding_mask=None,
                tgt_reference_points=ref, memory_text=text, text_attention_mask=tmask, memory=memory,
                memory_key_padding_mask=None, memory_level_start_index=start, memory_spatial_shapes=shapes,
                memory_pos=None, self_attn_mask=None, cross_attn_mask=None)
    out = out[0] if isinstance(out, tuple) else out
    torch.autograd.grad(out, [tgt, memory_bf, text], grad_outputs=g)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda r: -r.self_device_time_total)
rows = [r for r in rows if r.self_device_time_total > 0]
tot = sum(r.self_device_time_total for r in rows)
print("total device time %.1f us over %d kernels" % (tot, sum(r.count for r in rows)))
for r in rows[:60]:
    print("  %8.1f us x%-3d %s" % (r.self_device_time_total, r.count, r.key[:120]))
