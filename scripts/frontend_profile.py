#!/usr/bin/env python3
"""Dev: kernel time of the frozen front end (Swin-T + BERT) launched eagerly, for
  rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 scripts/frontend_profile.py [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ziragroundingdino_amd.config import zira_swint_config  # noqa: E402
from ziragroundingdino_amd.groundingdino import build_model  # noqa: E402
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda")
torch.manual_seed(0)
model = build_model(zira_swint_config(device="cuda")).to(dev).train()
model.use_frontend_graphs = False
ZiraTrainer(model)        # freezes what the task freezes (the frozen linears pick their arithmetic by requires_grad)
data = synthetic_batch(2, 800, 1333, device=dev)
which = os.environ.get("PART", "both")
if os.environ.get("GEMM_ARITH"):      # f32 | bf16x3
    from ziragroundingdino_amd import transformer as _zt
    _zt.Switches.gemm_arith = os.environ["GEMM_ARITH"]
from ziragroundingdino_amd.utils import nested_tensor_from_tensor_list  # noqa: E402
with torch.no_grad():
    samples = nested_tensor_from_tensor_list(model.preprocess_image(data))
    captions, _ = model._captions(data)
    for _ in range(iters + 2):
        if which in ("both", "swin"):
            model.run_backbone(samples)
        if which in ("both", "bert"):
            model.encode_text(captions, samples.device)
torch.cuda.synchronize()
