#!/usr/bin/env python3
"""Golden vectors for the task chain: two tasks trained one after the
other with the REFERENCE's modules on the CPU, re-enacting what its driver does around the step
(train_multidatasets.py):

    per task   load_model(): a freshly constructed model + load_state_dict(previous
               model_final.pth, strict=False)                                   :324-331, :559
               before_train(): freeze all but "adapter"                         :239-246
               max_iter x [forward, sum, backward, clip 0.1, AdamW, LR multiplier step]  :150-200, :447
               after_train(): __rep__ merge of every side branch                :221-237
               model_final.pth <- state_dict after the merge                    :319-322

The LR multiplier is detectron2's ``LRMultiplier`` (a ``LambdaLR`` stepped after every iteration)
over ``modified_coco_scheduler`` (coco_schedule.py:91-125): 1.0 until the decay iteration, then 0.1.
Same shrunken slice and name-seeded start as gen_step_golden.py; the second task starts from
default-initialised modules so that everything it uses must come through the checkpoint.

    python tests/golden/gen_tasks_golden.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
from gen_step_golden import CFG, SALT, SCALES, Slice, build_inputs  # noqa: E402

TASKS = [dict(name="aquarium", categories=["fish", "jellyfish", "shark"], seed=31, max_iter=3, decay_iter=2,
              input_ids=[101, 3000, 1012, 3001, 3002, 1012, 3003, 1012, 102]),
         dict(name="pothole", categories=["pothole", "crack", "manhole cover"], seed=32, max_iter=2, decay_iter=1,
              input_ids=[101, 3005, 1012, 3006, 1012, 3007, 3008, 1012, 102])]


def main():
    torch.set_num_threads(1)
    ref = ref_import.load()
    checkpoint, out = None, []
    for k, task in enumerate(TASKS):
        torch.manual_seed(100 + k)                     # default init of the fresh modules of task 2
        S = Slice(ref, seeded=checkpoint is None)
        if checkpoint is not None:
            S.load_state_dict(checkpoint)
        inp = build_inputs(torch.Generator().manual_seed(task["seed"]))
        inp["input_ids"] = torch.tensor([task["input_ids"]] * 2)
        opt = S.optimizer()
        sched = torch.optim.lr_scheduler.LambdaLR(
            opt, lambda it, t=task: 1.0 if it < t["decay_iter"] else 0.1)
        totals, lrs = [], []
        for it in range(task["max_iter"]):
            loss_dict = S.forward(inp)
            total = sum(loss_dict.values())
            opt.zero_grad()
            total.backward()
            torch.nn.utils.clip_grad_norm_([p for _, p in S.named_trainable() if p.grad is not None],
                                           max_norm=0.1, norm_type=2)
            lrs.append(max(g["lr"] for g in opt.param_groups))
            opt.step()
            sched.step()
            totals.append(total.detach().clone())
        before_rep = {n: v.clone() for n, v in S.state_dict().items() if n.startswith("rep_linear_adapter.")}
        S.rep()
        checkpoint = {n: v.clone() for n, v in S.state_dict().items()}
        out.append(dict(task=task, inputs=inp, totals=totals, lrs=lrs, before_rep=before_rep,
                        merged={n: v for n, v in checkpoint.items() if "adapter" in n}))
    path = os.path.join(HERE, "tasks_zira_slice.pt")
    torch.save(dict(cfg=CFG, salt=SALT, scales=SCALES, tasks=out), path)
    print("tasks_zira_slice %.1f KiB" % (os.path.getsize(path) / 1024))
    for o in out:
        print(o["task"]["name"], "totals", [round(float(t), 5) for t in o["totals"]], "lrs", o["lrs"])


if __name__ == "__main__":
    main()
