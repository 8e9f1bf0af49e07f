#!/usr/bin/env python3
"""The step golden of gen_step_golden.py at a size where this package's FROZEN-WEIGHT nodes accept the call.

step_zira_slice.pt is too small for them (dim_feedforward 64, 2 sampling points, 108 image tokens, a padded image):
the one-node decoder layer wants d_ffn % 128 == 0 and 3 * heads * levels * points % 128 == 0 (decoder_layer.py
``applies``), the encoder's attention node and the frozen FFN + LayerNorm node want unpadded images and at least 8192
token rows (the row LayerNorm kernel, dense.layer_norm_supported).  Here: dim_feedforward 128, 4 points, levels of
64 x 64 / 32 x 32 / 16 x 16 (+ 8 x 8 from the stride-2 projection) = 5440 tokens per image, two unpadded images, 32
queries -- the REFERENCE's modules on the CPU, two optimisation steps, exactly as gen_step_golden.py runs them
(reference transformer_for_adapter.py:910-1073, :809-907 under the freeze of groundingdino_dual_zero_rep_branch.py:722-745).
Inputs are regenerated from the seed by the test (``native_inputs``), the fixture holds results only.

    python tests/golden/gen_step_native_golden.py      (needs /root/reference; never runs on the GPU box)
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_step_golden as base  # noqa: E402

NATIVE = dict(dim_feedforward=128, enc_n_points=4, dec_n_points=4, num_queries=32)
SHAPES = ((64, 64), (32, 32), (16, 16))
IMAGE = (512, 512)
SEED = 23


def native_cfg():
    return dict(base.CFG, **NATIVE)


def native_inputs():
    """The minibatch of this fixture (the test calls this too)."""
    old = dict(base.CFG)
    base.CFG.update(NATIVE)
    try:
        return base.build_inputs(torch.Generator().manual_seed(SEED), shapes=SHAPES, image=IMAGE, pad_from=None)
    finally:
        base.CFG.clear()
        base.CFG.update(old)


def main():
    import ref_import

    torch.set_num_threads(os.cpu_count() or 1)
    inp = native_inputs()
    base.CFG.update(NATIVE)
    S = base.Slice(ref_import.load())
    named = S.named_trainable()
    opt = S.optimizer()
    steps = []
    for it in range(2):
        loss_dict = S.forward(inp)
        total = sum(loss_dict.values())
        opt.zero_grad()
        total.backward()
        grads = {n: p.grad.detach().clone() for n, p in named}
        gnorm = torch.nn.utils.clip_grad_norm_([p for _, p in named if p.grad is not None], max_norm=0.1, norm_type=2)
        opt.step()
        steps.append(dict(loss_dict={k: v.detach().clone() for k, v in loss_dict.items()}, total=total.detach().clone(),
                          grad_norm=gnorm.detach().clone(), grads=grads if it == 0 else None,
                          params_after={n: p.detach().clone() for n, p in named} if it == 1 else None))
    path = os.path.join(HERE, "step_zira_slice_native.pt")
    torch.save(dict(cfg=native_cfg(), salt=base.SALT, scales=base.SCALES, steps=steps,
                    trainable_names=[n for n, _ in named]), path)
    print("step_zira_slice_native %.1f KiB; losses step0:" % (os.path.getsize(path) / 1024),
          {k: round(float(v), 5) for k, v in steps[0]["loss_dict"].items()})
    print("grad norm", [float(s["grad_norm"]) for s in steps])


if __name__ == "__main__":
    main()
