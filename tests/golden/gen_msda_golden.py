#!/usr/bin/env python3
"""Generate golden vectors for the MSDA op FROM THE REFERENCE ITSELF.

Runs only in the build container (it needs /root/reference); the resulting
``msda_*.npz`` files are committed and are what travels to the GPU box.

It imports the reference's ``ms_deform_attn.py`` by file path with an empty stub for
the (CUDA-only, unbuilt) ``groundingdino._C`` extension and evaluates
``multi_scale_deformable_attn_pytorch`` (ms_deform_attn.py:90-130) plus its autograd --
the reference's own CPU path for this op -- on seeded inputs.

    python tests/golden/gen_msda_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference_msda():
    sys.path.insert(0, REF)
    import groundingdino  # noqa: F401  (empty __init__)

    stub = types.ModuleType("groundingdino._C")
    sys.modules["groundingdino._C"] = stub
    groundingdino._C = stub
    spec = importlib.util.spec_from_file_location(
        "ref_ms_deform_attn",
        os.path.join(REF, "groundingdino/models/GroundingDINO/ms_deform_attn.py"),
    )
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def level_start(shapes):
    hw = [h * w for h, w in shapes]
    return [0] + list(np.cumsum(hw)[:-1])


def make_case(name, B, Q, M, D, shapes, P, seed, dtype=torch.float32, loc_mode="unit"):
    g = torch.Generator().manual_seed(seed)
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    value = torch.randn(B, S, M, D, generator=g, dtype=dtype)
    attn = torch.randn(B, Q, M, L * P, generator=g, dtype=dtype).softmax(-1).view(B, Q, M, L, P)
    if loc_mode == "unit":          # U(0,1)
        loc = torch.rand(B, Q, M, L, P, 2, generator=g, dtype=dtype)
    elif loc_mode == "oob":         # U(-0.25, 1.25): many samples partly/fully outside
        loc = torch.rand(B, Q, M, L, P, 2, generator=g, dtype=dtype) * 1.5 - 0.25
    elif loc_mode == "grid":
        # exactly representable pixel centres / pixel borders (power-of-two maps so the
        # reference's (2*loc-1) -> ((g+1)*W-1)/2 and the kernel's loc*W-0.5 are both exact),
        # including -0.5/W (h_im == -1: excluded by the strict guard) and 1+0.5/W.
        loc = torch.empty(B, Q, M, L, P, 2, dtype=dtype)
        for l, (h, w) in enumerate(shapes):
            kx = torch.randint(-1, 2 * w + 2, (B, Q, M, P), generator=g)
            ky = torch.randint(-1, 2 * h + 2, (B, Q, M, P), generator=g)
            loc[:, :, :, l, :, 0] = kx.to(dtype) * 0.5 / w
            loc[:, :, :, l, :, 1] = ky.to(dtype) * 0.5 / h
    else:
        raise ValueError(loc_mode)
    grad_out = torch.randn(B, Q, M * D, generator=g, dtype=dtype)
    return dict(name=name, value=value, shapes=shapes, loc=loc, attn=attn, grad_out=grad_out)


CASES = [
    # BASELINE.json configs[0]: bs=1, Lq=100, 1 level, 4 heads
    dict(name="cfg0_b1_q100_l1_m4", B=1, Q=100, M=4, D=32, shapes=[(16, 20)], P=4, seed=1),
    dict(name="multi_b2_q37_m8_d32", B=2, Q=37, M=8, D=32,
         shapes=[(9, 13), (5, 7), (3, 4), (2, 2)], P=4, seed=2),
    dict(name="oob_b2_q19_m2_d16_p3", B=2, Q=19, M=2, D=16,
         shapes=[(5, 7), (3, 4), (1, 3)], P=3, seed=3, loc_mode="oob"),
    dict(name="grid_b1_q41_m2_d32", B=1, Q=41, M=2, D=32,
         shapes=[(8, 16), (4, 4), (1, 2), (2, 1)], P=4, seed=4, loc_mode="grid"),
    dict(name="d64_b1_q23_m3_p2", B=1, Q=23, M=3, D=64, shapes=[(6, 5), (3, 3)], P=2, seed=5,
         loc_mode="oob"),
    dict(name="d8_b3_q11_m5_p1", B=3, Q=11, M=5, D=8, shapes=[(7, 9)], P=1, seed=6,
         loc_mode="oob"),
    dict(name="d20_b1_q13_m2_p5", B=1, Q=13, M=2, D=20, shapes=[(4, 6), (3, 2), (2, 2)], P=5,
         seed=7, loc_mode="oob"),
    dict(name="f64_b2_q17_m4_d32", B=2, Q=17, M=4, D=32,
         shapes=[(9, 13), (5, 7), (3, 4), (2, 2)], P=4, seed=8, dtype=torch.float64,
         loc_mode="oob"),
]


def main():
    ref = load_reference_msda()
    for spec in CASES:
        c = make_case(**spec)
        shapes_t = torch.tensor(c["shapes"], dtype=torch.long)
        value = c["value"].clone().requires_grad_(True)
        loc = c["loc"].clone().requires_grad_(True)
        attn = c["attn"].clone().requires_grad_(True)
        out = ref.multi_scale_deformable_attn_pytorch(value, shapes_t, loc, attn)
        gv, gl, ga = torch.autograd.grad(out, (value, loc, attn), c["grad_out"])
        path = os.path.join(HERE, "msda_%s.npz" % c["name"])
        np.savez_compressed(
            path,
            value=c["value"].numpy(), spatial_shapes=shapes_t.numpy(),
            level_start_index=np.asarray(level_start(c["shapes"]), dtype=np.int64),
            sampling_loc=c["loc"].numpy(), attn_weight=c["attn"].numpy(),
            grad_output=c["grad_out"].numpy(),
            output=out.detach().numpy(), grad_value=gv.numpy(),
            grad_sampling_loc=gl.numpy(), grad_attn_weight=ga.numpy(),
        )
        print("%-28s out %s  %6.1f KiB" % (c["name"], tuple(out.shape), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    torch.set_num_threads(1)
    main()
