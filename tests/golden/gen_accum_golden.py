#!/usr/bin/env python3
"""Golden vectors for ``batch_size_scale`` gradient accumulation: three iterations of the shrunken ZiRa
slice (gen_step_golden.py: the REFERENCE's modules on the CPU) under the step rule of the reference's
``Trainer.run_step`` (train_multidatasets.py:192-199, non-AMP branch) with batch_size_scale = 2:

    losses.backward(); clip_grad_norm_(0.1) on the accumulated .grad EVERY iteration;
    if iter % 2 == 0: optimizer.step(); optimizer.zero_grad()          (iterations 0 and 2 step)

The iterations alternate between two minibatches.  The reference's Trainer class itself cannot be
imported here (detectron2's SimpleTrainer is not installed); those five lines are re-enacted.
      python tests/golden/gen_accum_golden.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
from gen_step_golden import CFG, SALT, SCALES, Slice, build_inputs  # noqa: E402

BATCH_SIZE_SCALE = 2


def main():
    torch.set_num_threads(1)
    S = Slice(ref_import.load())
    inputs = [build_inputs(torch.Generator().manual_seed(17)), build_inputs(torch.Generator().manual_seed(29))]
    named = S.named_trainable()
    opt = S.optimizer()
    opt.zero_grad()
    totals, norms = [], []
    for it in range(3):
        loss_dict = S.forward(inputs[it % 2])
        total = sum(loss_dict.values())
        total.backward()                                   # accumulates
        params = [p for _, p in named if p.grad is not None]
        norms.append(torch.nn.utils.clip_grad_norm_(params, max_norm=0.1, norm_type=2).detach().clone())
        if it % BATCH_SIZE_SCALE == 0:
            opt.step()
            opt.zero_grad()
        totals.append(total.detach().clone())
    path = os.path.join(HERE, "accum_zira_slice.pt")
    torch.save(dict(cfg=CFG, salt=SALT, scales=SCALES, inputs=inputs, batch_size_scale=BATCH_SIZE_SCALE,
                    totals=totals, grad_norms=norms, params_after={n: p.detach().clone() for n, p in named},
                    trainable_names=[n for n, _ in named]), path)
    print("accum_zira_slice %.1f KiB; totals" % (os.path.getsize(path) / 1024), [float(t) for t in totals],
          "norms", [float(n) for n in norms])


if __name__ == "__main__":
    main()
