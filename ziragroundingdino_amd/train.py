"""Training-step harness: the part of the reference's driver that touches the hot path
(train_multidatasets.py ``Trainer.run_step`` :150-200, ``before_train/after_train`` :221-246,
optimizer / DDP set-up in ``do_train`` :392-409, task config
test_odinw13_softfreeze/for_train/test_aquarium.py:13-25).

    forward -> sum(loss_dict) -> backward -> [all-reduce] -> clip_grad_norm_(0.1, L2) -> AdamW

``batch_size_scale`` = k (train_multidatasets.py:129-130, :192-199): gradients pile up in ``.grad`` over k
iterations, the clip runs on the piled-up gradients EVERY iteration, the optimizer steps (and the gradients
are cleared) only when ``iter % k == 0`` -- iteration 0 included.  fp16 autocast takes the reference's
GradScaler branch (:185-191).

Data parallelism, MI355X-first: one process per GPU; only the ZiRa side branches ever receive
gradients (4.6 M values, 18.5 MB fp32), so their ``.grad`` tensors are views of ONE flat buffer
that is all-reduced over RCCL/xGMI in a single call right after backward (the side branches are
the earliest layers, so their gradients are the last thing backward produces -- there is
nothing left to overlap with) and clipped from the same buffer.  The reference wraps the whole
170 M-parameter model in DDP before freezing it (``find_unused_parameters=True``) and
all-reduces every bucket, ~690 MB per step, for the same result.
"""
import math
from typing import Dict, List

import torch
import torch.distributed as dist

from .structures import Boxes, Instances


def lr_factor(name: str) -> float:
    """``optimizer.params.lr_factor_func`` of the task configs: twins learn at 0.2x."""
    return 0.2 if "freeze" in name else 1.0


class ZiraTrainer:
    def __init__(self, model, lr=1e-3, weight_decay=1e-4, betas=(0.9, 0.999), clip_max_norm=0.1,
                 clip_norm_type=2.0, process_group=None, tuned_gemms=True, amp_dtype=None, batch_size_scale=1,
                 grad_scaler=None):
        self.model = model
        # ``train.amp.enabled`` of the reference's configs (Trainer.run_step :170-174 wraps the forward in
        # autocast); here the dtype is named.  bf16 needs no GradScaler; fp16 (the reference's autocast default)
        # gets one, as in Trainer.__init__ :131-136.  The native fp32 ops (MSDA, side branch epilogue,
        # LayerNorm) keep computing in fp32 inside the autocast region.
        assert amp_dtype in (None, torch.bfloat16, torch.float16), "amp_dtype: None, torch.bfloat16 or torch.float16"
        self.amp_dtype = amp_dtype
        self.grad_scaler = grad_scaler
        if amp_dtype is torch.float16 and grad_scaler is None:
            self.grad_scaler = torch.amp.GradScaler(next(model.parameters()).device.type)
        assert int(batch_size_scale) >= 1
        self.batch_size_scale = int(batch_size_scale)
        # the reference unscales + clips every iteration but steps every k-th (train_multidatasets.py:185-199): with a
        # GradScaler that is "unscale_() has already been called" on the second iteration -- a latent bug there; refused here
        assert not (amp_dtype is torch.float16 and self.batch_size_scale > 1), \
            "fp16 (GradScaler) cannot be combined with batch_size_scale > 1: use bf16 or fp32 for accumulated steps"
        if tuned_gemms and next(model.parameters()).is_cuda:
            from . import tuned_gemm

            tuned_gemm.enable()  # per-shape GEMM kernel choices recorded for this step (see tuned_gemm.py)
        self._lr, self._wd, self._betas = lr, weight_decay, betas
        self.clip_max_norm, self.clip_norm_type = clip_max_norm, clip_norm_type
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.iter = 0
        self.always_reduce = False  # tests: issue the collective on a one-rank group too
        self.on_reduced_grad = None  # tests: callable(flat_grad) right after the all-reduce, before clip and step
        self._prefetched = None     # front end of the next minibatch, queued by run_step(..., next_data=)
        if hasattr(model, "criterion") and hasattr(model.criterion, "process_group"):
            model.criterion.process_group = process_group  # num_boxes is averaged over the same ranks
        self._bind()

    def _bind(self):
        """(Re)build the flat gradient bucket and the optimizer over the model's CURRENT trainable
        tensors.  Called at construction and again after ``after_train``: the re-parameterisation
        replaces every ``scaling`` parameter with a new tensor (as the reference does), which the old
        bucket and optimizer would no longer train."""
        model, lr, weight_decay, betas = self.model, self._lr, self._wd, self._betas
        model.before_train()  # freeze everything but the side branches
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        # one flat gradient bucket; .grad of every trainable tensor is a view into it
        total = sum(p.numel() for p in self.params)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=self.params[0].device)
        off = 0
        for p in self.params:
            p.grad = self.flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()
        # one parameter group per learning rate (the reference builds one per tensor; AdamW is
        # element-wise, so the result is the same and the multi-tensor kernels see 12-13 tensors
        # each instead of 25 groups of one)
        by_lr = {}
        for n, p in named:
            by_lr.setdefault(lr * lr_factor(n), []).append(p)
        groups = [{"params": ps, "lr": g_lr, "weight_decay": weight_decay} for g_lr, ps in by_lr.items()]
        # fused = one launch per parameter group on the GPU instead of ~10 multi-tensor launches (clip + step 0.8 -> 0.4 ms);
        # the same update rule, element-wise
        fused = self.params[0].is_cuda and self.fused_optimizer
        self.optimizer = torch.optim.AdamW(groups, lr=lr, betas=betas, weight_decay=weight_decay, fused=fused)

    fused_optimizer = True   # class-level switch (tests compare with the multi-tensor implementation)

    def _check_bucket(self):
        """The all-reduce, the clipping and the zeroing act on the flat bucket only: a ``.grad`` that no
        longer lives in it (``zero_grad(set_to_none=True)``, a replaced parameter) would silently train on
        raw local gradients."""
        lo = self.flat_grad.data_ptr()
        hi = lo + self.flat_grad.numel() * self.flat_grad.element_size()
        for n, p in zip(self.names, self.params):
            if p.grad is None or not (lo <= p.grad.data_ptr() < hi):
                raise RuntimeError("[ZiraTrainer] .grad of %s left the flat gradient bucket "
                                   "(zero_grad(set_to_none=True)?); call trainer._bind()" % n)

    prefetch_after_encoder = True   # class-level switch for A/B runs (False: behind the whole forward, as until round 5)
    # Developer switch (scripts/repro_frontend_hang.py): queue it BEFORE this step's forward.  Slower (26.4 against 25.4 ms), and
    # with the library's fp32 GEMMs for the large frozen products it HANGS THE GPU: every fp32 GEMM hipBLASLt / rocBLAS pick
    # on gfx950 is a Stream-K kernel (persistent grid, workgroups wait for their peers' partial sums), and two streams that
    # both run large ones at the same time deadlock (scripts/repro_streamk_two_streams.py: no model needed).  Behind the
    # encoder no large library GEMM of the step runs beside the front end, whatever the arithmetic.
    prefetch_at_start = False

    def run_step(self, data, next_data=None) -> Dict[str, torch.Tensor]:
        """One optimisation step on one minibatch; returns the (detached) weighted loss dict.
        ``next_data``: the minibatch of the NEXT step, if the caller has it already (a data loader with prefetch
        does): its frozen front end is queued on a second stream as soon as this step's forward has been launched,
        and picked up by the next ``run_step(next_data, ...)`` (see GroundingDINO.prefetch_frontend)."""
        assert self.model.training, "[ZiraTrainer] model was changed to eval mode!"
        self._check_bucket()
        kw = {}
        pre = self._prefetched
        self._prefetched = None
        if pre is not None and pre["inputs"] is data:
            kw["frontend"] = pre
        can_prefetch = (next_data is not None and self.amp_dtype is None and hasattr(self.model, "can_prefetch_frontend")
                        and self.model.can_prefetch_frontend())
        tr = getattr(self.model, "transformer", None)
        if can_prefetch and self.prefetch_at_start:
            from .transformer import Switches
            if Switches.gemm_arith == "f32" and not getattr(self, "allow_deadlock_repro", False):
                raise RuntimeError("[ZiraTrainer] prefetch_at_start with gemm_arith = 'f32' deadlocks the GPU (two streams of "
                                   "Stream-K library GEMMs); see scripts/repro_streamk_two_streams.py")
            self._prefetched = self.model.prefetch_frontend(next_data)
        elif can_prefetch and self.prefetch_after_encoder and hasattr(tr, "fire_after_encoder"):
            # the next minibatch's front end is queued when this step's ENCODER forward has been launched (round 5: 34.1 ->
            # 32.4 ms per step against queuing it behind the whole forward; queuing it at the start of the step hangs the GPU)
            def _queue(self=self, next_data=next_data):
                self._prefetched = self.model.prefetch_frontend(next_data)
            tr.__dict__["after_encoder"] = _queue
        try:
            if self.amp_dtype is not None:
                # (graph capture under autocast needs the weight-cast cache off: torch.cuda.make_graphed_callables)
                graphs = bool(getattr(self.model, "use_transformer_graph", False))
                with torch.autocast(self.flat_grad.device.type, dtype=self.amp_dtype, cache_enabled=not graphs):
                    loss_dict = self.model(data, **kw)
            else:
                loss_dict = self.model(data, **kw)
        finally:
            if tr is not None:   # (not fired: pieces still being captured, a forward that skipped it -- or one that raised: the
                tr.__dict__["after_encoder"] = None   # hook must not stay armed, holding the next minibatch)
        if can_prefetch and self._prefetched is None:
            self._prefetched = self.model.prefetch_frontend(next_data)
        losses = getattr(loss_dict, "total", None)   # the model's own sum of the same terms (criterion.LossDict)
        if losses is None:
            losses = sum(loss_dict.values())
        scaler = self.grad_scaler if self.amp_dtype is torch.float16 else None
        (scaler.scale(losses) if scaler is not None else losses).backward()
        if self.world > 1 or self.always_reduce:  # single RCCL all-reduce of the side-branch gradients
            # (with batch_size_scale > 1 the bucket also holds the earlier iterations' gradients: they are
            # identical on every rank, so sum / world leaves them as they are -- what DDP's reducer does too)
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            self.flat_grad.div_(self.world)
        if self.on_reduced_grad is not None:
            self.on_reduced_grad(self.flat_grad)
        if self.clip_max_norm is not None:
            if scaler is not None:
                scaler.unscale_(self.optimizer)   # (:187-189; without a clip, scaler.step() unscales)
            # clip_grad_norm_ over the side-branch tensors == one norm of the flat bucket; every iteration (:188-189, :194-195)
            total_norm = torch.linalg.vector_norm(self.flat_grad, self.clip_norm_type)
            self.flat_grad.mul_(torch.clamp(self.clip_max_norm / (total_norm + 1e-6), max=1.0))
        if self.iter % self.batch_size_scale == 0:   # (:190, :196)
            if scaler is not None:
                scaler.step(self.optimizer)
                scaler.update()
            else:
                self.optimizer.step()
            self.flat_grad.zero_()  # keeps the views alive (no set_to_none)
        self.iter += 1
        return {k: v.detach() for k, v in loss_dict.items()}

    def after_train(self, class_names: List[str] = ()):
        """End of a task (reference Trainer.after_train :221-237)."""
        self.model.add_cls_prompt(list(class_names))
        self.model.after_train()
        self._bind()  # __rep__ created new `scaling` parameters: train those in the next task


def synthetic_batch(batch_size, height=800, width=1333, n_categories=7, boxes_per_image=5, seed=0,
                    device="cpu"):
    """Synthetic minibatch in the reference's input format: uint8-valued
    images, one caption listing all categories, ``boxes_per_image`` xyxy boxes with side
    5-40 % of the image."""
    g = torch.Generator().manual_seed(seed)
    words = ["fish", "jellyfish", "penguin", "puffin", "shark", "starfish", "stingray", "crab",
             "turtle", "seal", "whale", "coral", "eel"]
    names = [words[i % len(words)] + ("" if i < len(words) else str(i)) for i in range(n_categories)]
    caption = " . ".join(names) + " ."
    batch = []
    for _ in range(batch_size):
        img = torch.randint(0, 256, (3, height, width), generator=g, dtype=torch.uint8).float()
        wh = (0.05 + 0.35 * torch.rand(boxes_per_image, 2, generator=g))
        c = wh / 2 + (1 - wh) * torch.rand(boxes_per_image, 2, generator=g)
        xyxy = torch.cat([c - wh / 2, c + wh / 2], -1) * torch.tensor([width, height, width, height])
        inst = Instances((height, width), gt_boxes=Boxes(xyxy),
                         gt_classes=torch.randint(0, n_categories, (boxes_per_image,), generator=g))
        batch.append({"image": img.to(device), "captions": caption, "instances": inst.to(device),
                      "height": height, "width": width})
    return batch
