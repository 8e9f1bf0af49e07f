#!/usr/bin/env python3
"""Dev: same-box A/B of class-level switches on the replayed training step.  `python scripts/ab_step.py NAME=0|1 ...` with
NAME in {dummy_launches (N tiny adds per fusion block), gemm_arith (f32|bf16x3), shared_source, batch_value, native_layer, native_glue, native_attention, overlap_text, compose_text, residual_in_gemm, graph_encoder, graph_decoder}: sets the switch, times 40 steps after 8 warm-up steps (graph replay, 4 rotating minibatches),
prints ms per step.  Run the variants alternately in ONE gpurun call, several times each: processes on one box differ by up to
0.5 ms; two trainers in one process do not work as an A/B (the second one built is 3 ms slower whatever its switches)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd import transformer as zt  # noqa: E402
from ziragroundingdino_amd.config import zira_swint_config  # noqa: E402
from ziragroundingdino_amd.groundingdino import build_model  # noqa: E402
from ziragroundingdino_amd.ms_deform_attn import MultiScaleDeformableAttention as M  # noqa: E402
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch  # noqa: E402

n_categories = 15
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    if k == "categories":      # text tokens = 2 + 2 * categories (15 -> 32 ODinW-like, 96 -> 194 COCO-like)
        n_categories = int(v)
    elif k == "shared_source":
        M.fuse_shared_source = bool(int(v))
    elif k == "batch_value":
        zt.TransformerDecoder.batch_value_projections = bool(int(v))
    elif k == "native_layer":
        zt.DeformableTransformerDecoderLayer.native_layer = bool(int(v))
    elif k == "native_glue":
        zt.TransformerDecoder.native_glue = bool(int(v))
    elif k == "native_attention":
        zt.DeformableTransformerEncoderLayer.native_attention = bool(int(v))
    elif k == "overlap_text":
        zt.TransformerEncoder.overlap_text_layer = bool(int(v))
    elif k == "compose_text":
        zt.BiMultiHeadAttention.compose_text_side = bool(int(v))
    elif k == "residual_in_gemm":
        zt.BiAttentionBlock.residual_in_gemm = bool(int(v))
    elif k == "gemm_arith":
        zt.Switches.gemm_arith = v
    elif k == "panel":
        from ziragroundingdino_amd import gemm_bf16x3 as _g3
        _g3.USE_PANEL = bool(int(v))
    elif k == "fused_image_side":
        zt.BiAttentionBlock.fused_image_side = bool(int(v))
    elif k == "swin_epilogues":
        from ziragroundingdino_amd import backbone as _bb
        _bb.FUSED_EPILOGUES = bool(int(v))
    elif k == "skip_text_layer":     # upper bound of what the text enhancer layers cost the step (results are wrong)
        if int(v):
            zt.TransformerEncoderLayer.forward = lambda self, src, src_mask=None, src_key_padding_mask=None, pos=None: src + 0.0 * self.norm2.weight.sum()
    elif k == "tall":
        from ziragroundingdino_amd import dense as _dense
        _dense.USE_TALL_BF16X3 = bool(int(v))
    elif k == "group_norm":
        from ziragroundingdino_amd import dense as _dense
        _dense.USE_GROUP_NORM = bool(int(v))
    elif k == "thin":
        from ziragroundingdino_amd import dense as _dense
        _dense.USE_THIN = bool(int(v))
    elif k == "prefetch_after_encoder":
        ZiraTrainer.prefetch_after_encoder = bool(int(v))
    elif k == "prefetch_at_start":   # (hangs with gemm_arith=f32: two streams of Stream-K library GEMMs, scripts/repro_streamk_two_streams.py)
        ZiraTrainer.prefetch_at_start = bool(int(v))
        import faulthandler
        faulthandler.dump_traceback_later(90, exit=True)     # (a hung GPU must not hold the box)
    elif k == "frontend_graphs":
        _fg = bool(int(v))
    elif k == "native_pos":
        from ziragroundingdino_amd.backbone import PositionEmbeddingSineHW
        PositionEmbeddingSineHW.native = bool(int(v))
    elif k == "native_text_side":
        zt.BiAttentionBlock.native_text_side = bool(int(v))
    elif k == "native_losses":
        from ziragroundingdino_amd.criterion import TwoStageCriterion
        TwoStageCriterion.native_losses = bool(int(v))
    elif hasattr(zt.Switches, k):      # any boolean of transformer.Switches (native_geometry, fused_attention, ...)
        setattr(zt.Switches, k, bool(int(v)))
    elif k == "dummy_launches":   # what a tiny launch costs the step: N extra [64, 256] adds per fusion block forward
        _n, _orig = int(v), zt.BiAttentionBlock.forward

        def _with_dummies(self, *a, _n=_n, _orig=_orig, **kw):
            scratch = self.__dict__.get("_scratch")
            if scratch is None:
                scratch = self.__dict__["_scratch"] = torch.zeros(64, 256, device="cuda")
            for _ in range(_n):
                scratch.add_(1.0)
            return _orig(self, *a, **kw)
        zt.BiAttentionBlock.forward = _with_dummies
    elif k in ("graph_encoder", "graph_decoder", "graph_fusion", "graph_selection"):
        from ziragroundingdino_amd.graphs import GraphedTransformer
        setattr(GraphedTransformer, k, bool(int(v)))
    else:
        raise SystemExit("unknown switch " + k)
dev = torch.device("cuda")
torch.manual_seed(0)
model = build_model(zira_swint_config(device="cuda")).to(dev).train()
if "_fg" in globals():
    model.use_frontend_graphs = _fg
trainer = ZiraTrainer(model)
if os.environ.get("ZIRA_AB_FREE_FRONTEND") == "1":    # what the concurrent front end costs the step: one minibatch, its front end computed once
    n_distinct = 1
    _orig_pf, _cache = model.prefetch_frontend, {}

    def _cached_prefetch(inputs):
        if id(inputs) not in _cache:
            _cache[id(inputs)] = _orig_pf(inputs)
            torch.cuda.synchronize()
        return _cache[id(inputs)]
    model.prefetch_frontend = _cached_prefetch
else:
    n_distinct = 4
batches = [synthetic_batch(2, 800, 1333, n_categories=n_categories, seed=i, device=dev) for i in range(n_distinct)] * (4 // n_distinct)
for i in range(8):
    trainer.run_step(batches[i % 4], next_data=batches[(i + 1) % 4])
torch.cuda.synchronize()
ts = []
for rep in range(2):
    t0 = time.perf_counter()
    for i in range(20):
        trainer.run_step(batches[i % 4], next_data=batches[(i + 1) % 4])
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 20 * 1e3)
print("%s: %.3f / %.3f ms per step" % (" ".join(sys.argv[1:]) or "defaults", ts[0], ts[1]), flush=True)
