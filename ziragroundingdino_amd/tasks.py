"""Task chain: the loop the reference's driver runs around the training step when it learns one
dataset after another (train_multidatasets.py ``main`` :470-565, ``do_train`` :367-468,
``load_model`` :324-331, ``PeriodicCheckpointer`` :319-322).

    for each task:
        model  <- build_model();  load_state_dict(previous model_final.pth ["model"], strict=False)
        ZiraTrainer(model)        before_train(): only the side branches train
        max_iter x run_step       LR multiplier stepped after every iteration
        after_train(categories)   prompt pool entries + __rep__ merge of every side branch
        <output_dir>/model_final.pth <- {"model": state_dict, "trainer": ..., "iteration": max_iter}

Checkpoints keep the reference's (detectron2 ``Checkpointer``) layout -- a dict with the model's state
dict under ``"model"``, periodic files ``model_{iteration:07d}.pth`` and a ``last_checkpoint`` text file
naming the newest -- and the reference's key names, so either side can continue from the other's files.
``model_final.pth`` is written AFTER the merge (the reference's checkpointer hook saves once more from
``after_train``), which is what makes the next task start from merged twins and 1e-8 branches.

Only what touches the hot path is here: evaluation, metric writers, EMA and the dataset registry of
the reference's driver are not.
"""
import os
from dataclasses import dataclass
from typing import Callable, Dict, Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist

from .train import ZiraTrainer


def clean_state_dict(state_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Drop the ``module.`` prefix a DDP-wrapped model puts on every key (util/utils.py:21-28)."""
    return {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}


def multistep_lr_multiplier(decay_iter: int, gamma: float = 0.1) -> Callable[[int], float]:
    """``modified_coco_scheduler(epochs, decay_epochs, base_steps)`` with no warm-up
    (coco_schedule.py:91-125): 1.0 for iterations < decay_iter, ``gamma`` from there on."""
    return lambda it: 1.0 if it < decay_iter else gamma


@dataclass
class TaskSpec:
    """One task config of the reference (test_odinw13_softfreeze/for_train/test_*.py)."""
    name: str
    categories_names: Sequence[str]
    data: Callable[[int], Iterable]          # start_iter -> iterator of minibatches (endless or >= max_iter long)
    max_iter: int
    output_dir: str
    lr: float = 1e-3
    weight_decay: float = 1e-4
    clip_max_norm: Optional[float] = 0.1
    clip_norm_type: float = 2.0
    lr_multiplier: Optional[Callable[[int], float]] = None   # default: x0.1 after 40 % (10 epochs, decay at 4)
    checkpoint_period: Optional[int] = None                  # default: max_iter (the configs' 10 epochs)
    batch_size_scale: int = 1                                # optimizer step every k iterations (train_multidatasets.py:192-199)

    def multiplier(self) -> Callable[[int], float]:
        return self.lr_multiplier or multistep_lr_multiplier((self.max_iter * 4) // 10)


def load_model(build_model: Callable[[], torch.nn.Module], checkpoint_path: Optional[str], device="cpu"):
    """Fresh model + the previous task's weights.  ``strict=False`` as in the reference: a checkpoint
    from before ZiRa has no side-branch keys, and prompt pool entries create themselves on load."""
    model = build_model()
    if checkpoint_path:
        checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        model.load_state_dict(clean_state_dict(checkpoint["model"]), strict=False)
    return model.to(device).train()       # the reference trains with the whole model in train mode (:330)


def _is_main(group) -> bool:
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank(group) == 0


def _barrier(group):
    if dist.is_available() and dist.is_initialized():
        dist.barrier(group=group)


def save_checkpoint(output_dir: str, name: str, model, trainer: ZiraTrainer, iteration: int) -> str:
    os.makedirs(output_dir, exist_ok=True)
    path = os.path.join(output_dir, name + ".pth")
    tmp = path + ".tmp"
    torch.save({"model": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                "trainer": {"iteration": iteration, "optimizer": trainer.optimizer.state_dict()},
                "iteration": iteration}, tmp)
    os.replace(tmp, path)               # a killed run never leaves a half-written file under the final name
    with open(os.path.join(output_dir, "last_checkpoint"), "w") as f:
        f.write(name + ".pth")
    return path


def _resume(spec: TaskSpec, model, trainer: ZiraTrainer) -> int:
    """``checkpointer.resume_or_load(resume=True)`` (:461-466): continue after the iteration the newest
    periodic checkpoint of THIS task stored.  Returns the iteration to start at."""
    marker = os.path.join(spec.output_dir, "last_checkpoint")
    if not os.path.exists(marker):
        return 0
    with open(marker) as f:
        path = os.path.join(spec.output_dir, f.read().strip())
    checkpoint = torch.load(path, map_location="cpu", weights_only=False)
    model.load_state_dict(clean_state_dict(checkpoint["model"]), strict=False)
    trainer._bind()
    trainer.optimizer.load_state_dict(checkpoint["trainer"]["optimizer"])
    return int(checkpoint["trainer"]["iteration"]) + 1


def _check_matching(model, process_group=None):
    """The device-side matcher records what the reference stops on at once (scipy's ValueError for an
    infeasible -- inf / NaN -- cost matrix, ``generalized_box_iou``'s xyxy assertion:
    matcher/matcher.py:147, util/box_ops.py:51-52) as device flags and goes on with dummy assignments.
    Read them back here so that a diverged run ends before a checkpoint is written or merged.  With several ranks the
    verdict is all-reduced (MAX) first: every rank raises together instead of one raising and the others hanging in
    their next collective until the RCCL timeout."""
    criterion = getattr(model, "criterion", None)
    matcher = getattr(criterion, "matcher", None)
    err = None
    if matcher is not None and hasattr(matcher, "check"):
        try:
            matcher.check()
        except (AssertionError, ValueError) as exc:
            err = exc
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dev = next(model.parameters()).device
        flag = torch.tensor([1 if err is not None else 0], dtype=torch.int32,
                            device=dev if dist.get_backend(process_group) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=process_group)
        if err is None and int(flag.item()):
            err = RuntimeError("device-side matching failed on another rank (infeasible cost matrix or boxes not in xyxy order)")
    if err is not None:
        raise err


def run_task(spec: TaskSpec, build_model, init_checkpoint: Optional[str], device="cpu", process_group=None,
             resume=False, on_step=None) -> str:
    """``do_train`` for one task; returns the path of its ``model_final.pth``."""
    final = os.path.join(spec.output_dir, "model_final.pth")
    if resume and os.path.exists(final):
        return final
    model = load_model(build_model, init_checkpoint, device)
    trainer = ZiraTrainer(model, lr=spec.lr, weight_decay=spec.weight_decay, clip_max_norm=spec.clip_max_norm,
                          clip_norm_type=spec.clip_norm_type, process_group=process_group,
                          batch_size_scale=spec.batch_size_scale)
    base_lrs = [g["lr"] for g in trainer.optimizer.param_groups]   # before a resume: a checkpoint stores the
    start_iter = _resume(spec, model, trainer) if resume else 0    # MULTIPLIED rates of the iteration it was taken at
    trainer.iter = start_iter             # (the reference's step rule, iter % batch_size_scale == 0, counts from here)
    assert len(base_lrs) == len(trainer.optimizer.param_groups)
    multiplier = spec.multiplier()
    period = spec.checkpoint_period or spec.max_iter
    batches = iter(spec.data(start_iter))
    for it in range(start_iter, spec.max_iter):
        for g, base in zip(trainer.optimizer.param_groups, base_lrs):
            g["lr"] = base * multiplier(it)
        loss_dict = trainer.run_step(next(batches))
        if on_step is not None:
            on_step(spec, it, loss_dict)
        if (it + 1) % period == 0:
            _check_matching(model, process_group)        # before anything is written: one host sync per checkpoint period
            if _is_main(process_group):
                save_checkpoint(spec.output_dir, "model_%07d" % it, model, trainer, it)
    _check_matching(model, process_group)                # ... and before the side branches are merged into the weights
    trainer.after_train(list(spec.categories_names))
    if _is_main(process_group):
        save_checkpoint(spec.output_dir, "model_final", model, trainer, spec.max_iter)
    _barrier(process_group)               # every rank loads this file at the start of the next task
    return final


def run_tasks(specs: Sequence[TaskSpec], build_model, init_checkpoint: Optional[str] = None, device="cpu",
              process_group=None, resume=False, on_step=None) -> List[str]:
    """The task loop of ``main`` (:531-560): each task starts from the previous one's
    ``model_final.pth``.  Returns the list of final checkpoints, one per task."""
    finals = []
    for spec in specs:
        init_checkpoint = run_task(spec, build_model, init_checkpoint, device, process_group, resume, on_step)
        finals.append(init_checkpoint)
    return finals
