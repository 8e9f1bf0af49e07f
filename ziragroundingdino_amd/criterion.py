"""Set-prediction loss of GroundingDINO / ZiRa (reference
groundingdino/models/GroundingDINO/criterion/criterion.py:62-262 ``SetCriterion``,
two_stage_criterion.py:20-100 ``TwoStageCriterion``, criterion/__init__.py:22-40 weights).

Loss-dict keys are the reference's: ``loss_class / loss_bbox / loss_giou`` plus the suffixes
``_0.._{dec_layers-2}`` (auxiliary decoder layers) and ``_enc`` (two-stage encoder output).
"""
import copy
from typing import List

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from .box_ops import box_cxcywh_to_xyxy, generalized_box_iou, generalized_box_iou_aligned
from .matcher import build_matcher


def is_dist_avail_and_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def get_world_size() -> int:
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def sigmoid_focal_loss(inputs, targets, num_boxes, alpha: float = 0.25, gamma: float = 2):
    """RetinaNet focal loss, mean over queries then sum, / num_boxes (reference criterion.py:31-59)."""
    prob = inputs.sigmoid()
    ce_loss = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = prob * targets + (1 - prob) * (1 - targets)
    loss = ce_loss * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.mean(1).sum() / num_boxes


class LossDict(dict):
    """A plain loss dict that also carries how its entries were made: ``stacked[name] = (vector [S], suffixes)`` says
    that ``self[name + suffixes[s]]`` is ``vector[s]``.  A caller that only needs the weighted SUM can take it from the
    vectors (three products and sums) instead of from 21 scalar entries (a multiply, an add and a select-backward
    each).  Everything else treats it as the dict it is."""
    stacked = None
    total = None


class _StackedLosses(torch.autograd.Function):
    """Focal / L1 / GIoU losses of all prediction sets from the device-side matches: csrc/criterion.hip, two launches forward
    and one backward instead of ~60 + ~75 launch-bound ATen kernels (``_forward_stacked`` keeps the op chain for everything
    this does not take).  -> [3, S] = (loss_class, loss_bbox, loss_giou) per set."""

    @staticmethod
    def forward(ctx, logits, boxes, q_idx, t_idx, image_of, labels_all, boxes_all, num_boxes, alpha, gamma):
        from . import _lib
        S, B, Q, C = logits.shape
        M = q_idx.shape[1]
        lib = _lib.load()
        scratch = torch.empty(lib.zira_stacked_losses_scratch_bytes(S, B, Q, M), dtype=torch.uint8, device=logits.device)
        out = torch.empty((3, S), dtype=torch.float32, device=logits.device)
        logits, boxes = logits.contiguous(), boxes.contiguous()
        nb = num_boxes.reshape(1)
        with torch.cuda.device(logits.device):
            rc = lib.zira_stacked_losses_fwd_f32(logits.data_ptr(), boxes.data_ptr(), q_idx.data_ptr(), t_idx.data_ptr(),
                                                 image_of.data_ptr(), labels_all.data_ptr(), boxes_all.data_ptr(), nb.data_ptr(),
                                                 S, B, Q, C, M, float(alpha), float(gamma), scratch.data_ptr(), out.data_ptr(),
                                                 torch.cuda.current_stream(logits.device).cuda_stream)
        if rc != 0:
            raise RuntimeError("zira_stacked_losses_fwd_f32 failed: hipError %d" % rc)
        ctx.save_for_backward(logits, boxes, q_idx, t_idx, image_of, labels_all, boxes_all, nb)
        ctx.alpha, ctx.gamma = float(alpha), float(gamma)
        return out

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        logits, boxes, q_idx, t_idx, image_of, labels_all, boxes_all, nb = ctx.saved_tensors
        S, B, Q, C = logits.shape
        M = q_idx.shape[1]
        g = g.contiguous()
        g_logits = torch.empty_like(logits) if ctx.needs_input_grad[0] else None
        g_boxes = torch.empty_like(boxes) if ctx.needs_input_grad[1] else None
        if g_logits is not None or g_boxes is not None:
            with torch.cuda.device(logits.device):
                rc = _lib.load().zira_stacked_losses_bwd_f32(
                    logits.data_ptr(), boxes.data_ptr(), q_idx.data_ptr(), t_idx.data_ptr(), image_of.data_ptr(),
                    labels_all.data_ptr(), boxes_all.data_ptr(), nb.data_ptr(), g.data_ptr(), S, B, Q, C, M, ctx.alpha, ctx.gamma,
                    0 if g_logits is None else g_logits.data_ptr(), 0 if g_boxes is None else g_boxes.data_ptr(),
                    torch.cuda.current_stream(logits.device).cuda_stream)
            if rc != 0:
                raise RuntimeError("zira_stacked_losses_bwd_f32 failed: hipError %d" % rc)
        return g_logits, g_boxes, None, None, None, None, None, None, None, None


class SetCriterion(nn.Module):
    def __init__(self, num_classes, matcher, weight_dict, losses: List[str] = ["class", "boxes"],
                 eos_coef: float = 0.1, loss_class_type: str = "focal_loss", alpha: float = 0.25,
                 gamma: float = 2.0):
        super().__init__()
        assert loss_class_type in ["ce_loss", "focal_loss"]
        self.num_classes = num_classes
        self.matcher = matcher
        self.weight_dict = weight_dict
        self.losses = losses
        self.alpha = alpha
        self.gamma = gamma
        self.eos_coef = eos_coef
        self.loss_class_type = loss_class_type
        self.process_group = None  # ranks num_boxes is averaged over (None = the default group)
        if loss_class_type == "ce_loss":
            empty_weight = torch.ones(self.num_classes + 1)
            empty_weight[-1] = eos_coef
            self.register_buffer("empty_weight", empty_weight)

    @staticmethod
    def _get_src_permutation_idx(indices):
        batch_idx = torch.cat([torch.full_like(src, i) for i, (src, _) in enumerate(indices)])
        src_idx = torch.cat([src for (src, _) in indices])
        return batch_idx, src_idx

    def loss_labels(self, outputs, targets, indices, num_boxes):
        src_logits = outputs["pred_logits"]
        idx = self._get_src_permutation_idx(indices)
        target_classes_o = torch.cat([t["labels"][J] for t, (_, J) in zip(targets, indices)])
        target_classes = torch.full(src_logits.shape[:2], self.num_classes, dtype=torch.int64,
                                    device=src_logits.device)
        target_classes[idx] = target_classes_o
        if self.loss_class_type == "ce_loss":
            loss_class = F.cross_entropy(src_logits.transpose(1, 2), target_classes, self.empty_weight)
        else:
            onehot = torch.zeros([src_logits.shape[0], src_logits.shape[1], src_logits.shape[2] + 1],
                                 dtype=src_logits.dtype, device=src_logits.device)
            onehot.scatter_(2, target_classes.unsqueeze(-1), 1)
            loss_class = sigmoid_focal_loss(src_logits, onehot[:, :, :-1], num_boxes=num_boxes,
                                            alpha=self.alpha, gamma=self.gamma) * src_logits.shape[1]
        return {"loss_class": loss_class}

    def loss_boxes(self, outputs, targets, indices, num_boxes):
        idx = self._get_src_permutation_idx(indices)
        src_boxes = outputs["pred_boxes"][idx]
        target_boxes = torch.cat([t["boxes"][i] for t, (_, i) in zip(targets, indices)], dim=0)
        loss_bbox = F.l1_loss(src_boxes, target_boxes, reduction="none")
        loss_giou = 1 - torch.diag(generalized_box_iou(box_cxcywh_to_xyxy(src_boxes),
                                                       box_cxcywh_to_xyxy(target_boxes)))
        return {"loss_bbox": loss_bbox.sum() / num_boxes, "loss_giou": loss_giou.sum() / num_boxes}

    def get_loss(self, loss, outputs, targets, indices, num_boxes, **kwargs):
        loss_map = {"class": self.loss_labels, "boxes": self.loss_boxes}
        assert loss in loss_map, f"do you really want to compute {loss} loss?"
        return loss_map[loss](outputs, targets, indices, num_boxes, **kwargs)

    def _num_boxes(self, outputs, targets):
        num_boxes = sum(len(t["labels"]) for t in targets)
        num_boxes = torch.as_tensor([num_boxes], dtype=torch.float,
                                    device=next(iter(outputs.values())).device)
        world = 1
        if is_dist_avail_and_initialized():
            group = getattr(self, "process_group", None)  # set by ZiraTrainer: the ranks the gradients are averaged over
            dist.all_reduce(num_boxes, group=group)
            world = dist.get_world_size(group)
        # kept on the device (0-dim): the reference's .item() here is one more host sync per step
        return torch.clamp(num_boxes / world, min=1)[0]


class TwoStageCriterion(SetCriterion):
    def __init__(self, num_classes, matcher, weight_dict, losses=["class", "boxes"], eos_coef=None,
                 loss_class_type="focal_loss", alpha: float = 0.25, gamma: float = 2,
                 two_stage_binary_cls=False):
        super().__init__(num_classes, matcher, weight_dict, losses, eos_coef, loss_class_type,
                         alpha, gamma)
        self.two_stage_binary_cls = two_stage_binary_cls

    _index_cache = {}
    native_losses = True   # device-side matches: the focal / L1 / GIoU losses of all sets as one node (csrc/criterion.hip)

    @classmethod
    def _set_and_image_index(cls, S, sizes, Q, dev):
        """(set index, image index) of every matched pair in the device solver's [S, M] layout -- known from
        the shapes alone (min(Q, targets) pairs per image), built once per combination of target counts."""
        key = (S, tuple(sizes), Q, str(dev))
        hit = cls._index_cache.get(key)
        if hit is None:
            per_image = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor([min(n, Q) for n in sizes]))
            M = per_image.numel()
            s_i = torch.arange(S).repeat_interleave(M)
            if len(cls._index_cache) > 256:
                cls._index_cache.clear()
            hit = cls._index_cache[key] = (s_i.to(dev), per_image.repeat(S).to(dev))
        return hit

    def _forward_stacked(self, outputs, targets, return_indices):
        """All prediction sets of the step at once.  ``outputs["stacked"]`` = (logits [S, B, Q, C],
        boxes [S, B, Q, 4], suffixes) as the model's heads produce them: one cost computation and
        one device->host copy for the S*B matchings, one focal-loss pass, one L1 / GIoU pass --
        the values of the per-set loop below with 1/S of its ~600 kernel launches and 2 host syncs
        instead of 4 per set."""
        logits, boxes, suffixes = outputs["stacked"]
        S, B, Q, C = logits.shape
        dev = logits.device
        sizes = [len(t["labels"]) for t in targets]
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + n)
        if logits.is_cuda and hasattr(self.matcher, "forward_stacked_device"):
            # assignments solved on the device: no copy to the host, no synchronisation
            q_dev, t_dev = self.matcher.forward_stacked_device(logits, boxes, targets)
            s_i, b_i = self._set_and_image_index(S, sizes, Q, dev)
            q_i, t_i = q_dev.reshape(-1), t_dev.reshape(-1)
            counts = [q_dev.shape[1]] * S
            all_indices = None
        else:
            all_indices = self.matcher.forward_stacked(logits, boxes, targets)
            s_idx, b_idx, q_idx, t_idx, counts = [], [], [], [], []
            for s, per_image in enumerate(all_indices):
                n_s = 0
                for b, (src, tgt) in enumerate(per_image):
                    s_idx.append(torch.full_like(src, s))
                    b_idx.append(torch.full_like(src, b))
                    q_idx.append(src)
                    t_idx.append(tgt + offs[b])
                    n_s += len(src)
                counts.append(n_s)
            idx = torch.stack([torch.cat(s_idx), torch.cat(b_idx), torch.cat(q_idx), torch.cat(t_idx)]).to(dev)
            s_i, b_i, q_i, t_i = idx[0], idx[1], idx[2], idx[3]
        num_boxes = self._num_boxes({"pred_logits": logits}, targets)
        labels_all = torch.cat([t["labels"] for t in targets])
        boxes_all = torch.cat([t["boxes"] for t in targets])

        losses = LossDict()
        losses.stacked = {}
        if (self.native_losses and all_indices is None and "class" in self.losses and "boxes" in self.losses and q_dev.shape[1] > 0
                and logits.dtype == torch.float32 and boxes.dtype == torch.float32 and boxes_all.dtype == torch.float32
                and labels_all.dtype == torch.int64 and S <= 65535):
            # every loss of every set in two launches (csrc/criterion.hip)
            per = _StackedLosses.apply(logits, boxes, q_dev.contiguous(), t_dev.contiguous(), b_i[:q_dev.shape[1]].contiguous(),
                                       labels_all.contiguous(), boxes_all.contiguous(), num_boxes, self.alpha, self.gamma)
            for row, name in enumerate(("loss_class", "loss_bbox", "loss_giou")):
                losses.stacked[name] = (per[row], list(suffixes))
                for s, suf in enumerate(suffixes):
                    losses[name + suf] = per[row, s]
        elif "class" in self.losses:
            assert self.loss_class_type == "focal_loss"
            target_classes = torch.full((S, B, Q), self.num_classes, dtype=torch.int64, device=dev)
            target_classes[s_i, b_i, q_i] = labels_all[t_i]
            onehot = torch.zeros((S, B, Q, C + 1), dtype=logits.dtype, device=dev)
            onehot.scatter_(3, target_classes.unsqueeze(-1), 1)
            tgt = onehot[..., :-1]
            prob = logits.sigmoid()
            ce = F.binary_cross_entropy_with_logits(logits, tgt, reduction="none")
            p_t = prob * tgt + (1 - prob) * (1 - tgt)
            loss = ce * ((1 - p_t) ** self.gamma)
            if self.alpha >= 0:
                loss = (self.alpha * tgt + (1 - self.alpha) * (1 - tgt)) * loss
            per_set = loss.mean(2).sum((1, 2)) / num_boxes * Q
            losses.stacked["loss_class"] = (per_set, list(suffixes))
            for s, suf in enumerate(suffixes):
                losses["loss_class" + suf] = per_set[s]
        if "boxes" in self.losses and "loss_bbox" not in losses.stacked:
            src = boxes[s_i, b_i, q_i]
            tgt = boxes_all[t_i]
            l1 = F.l1_loss(src, tgt, reduction="none").sum(-1)
            giou = 1 - generalized_box_iou_aligned(box_cxcywh_to_xyxy(src), box_cxcywh_to_xyxy(tgt))
            if len(set(counts)) == 1 and counts[0] > 0:   # the usual case: every set matches every target
                l1_s = l1.view(S, -1).sum(1) / num_boxes
                giou_s = giou.view(S, -1).sum(1) / num_boxes
            else:
                seg = torch.zeros(S, dtype=l1.dtype, device=dev)
                l1_s = seg.index_add(0, s_i, l1) / num_boxes
                giou_s = seg.index_add(0, s_i, giou) / num_boxes
            losses.stacked["loss_bbox"] = (l1_s, list(suffixes))
            losses.stacked["loss_giou"] = (giou_s, list(suffixes))
            for s, suf in enumerate(suffixes):
                losses["loss_bbox" + suf] = l1_s[s]
                losses["loss_giou" + suf] = giou_s[s]
        if return_indices:
            if all_indices is None:   # device path: per set and image (query_idx, target_idx) views, still on the device
                moff = [0]
                for n in sizes:
                    moff.append(moff[-1] + min(n, Q))
                all_indices = [[(q_dev[s, moff[b]:moff[b + 1]], t_dev[s, moff[b]:moff[b + 1]] - offs[b])
                                for b in range(B)] for s in range(S)]
            by = dict(zip(suffixes, all_indices))
            return losses, {"indices": by.get(""), "aux_outputs": [by[k] for k in suffixes if k not in ("", "_enc")],
                            "enc_outputs": [by["_enc"]] if "_enc" in by else []}
        return losses

    def forward(self, outputs, targets, return_indices=False):
        if ("stacked" in outputs and hasattr(self.matcher, "forward_stacked")
                and self.loss_class_type == "focal_loss" and not self.two_stage_binary_cls):
            return self._forward_stacked(outputs, targets, return_indices)
        outputs_without_aux = {k: v for k, v in outputs.items()
                               if k not in ("aux_outputs", "enc_outputs", "cate_to_token_mask_list", "stacked")}
        aux_list = list(outputs.get("aux_outputs", []))
        enc_outputs = outputs.get("enc_outputs")
        if enc_outputs is not None and self.two_stage_binary_cls:
            raise NotImplementedError("two_stage_binary_cls is not used by the ZiRa configs")
        # all matchings of the step in one go (one device->host copy instead of one per set)
        sets = [outputs_without_aux] + aux_list + ([enc_outputs] if enc_outputs is not None else [])
        if hasattr(self.matcher, "forward_many"):
            all_indices = self.matcher.forward_many(sets, targets)
        else:
            all_indices = [self.matcher(o, targets) for o in sets]
        indices_list = {"indices": all_indices[0], "aux_outputs": all_indices[1:1 + len(aux_list)],
                        "enc_outputs": all_indices[1 + len(aux_list):]}
        num_boxes = self._num_boxes(outputs_without_aux, targets)

        losses = {}
        for loss in self.losses:
            losses.update(self.get_loss(loss, outputs, targets, all_indices[0], num_boxes))
        for i, aux_outputs in enumerate(aux_list):
            for loss in self.losses:
                l_dict = self.get_loss(loss, aux_outputs, targets, all_indices[1 + i], num_boxes)
                losses.update({k + f"_{i}": v for k, v in l_dict.items()})
        if enc_outputs is not None:
            for loss in self.losses:
                l_dict = self.get_loss(loss, enc_outputs, targets, all_indices[-1], num_boxes)
                losses.update({k + "_enc": v for k, v in l_dict.items()})
        if return_indices:
            return losses, indices_list
        return losses


def build_criterion(args):
    """class 1, bbox 5, giou 2, replicated for ``_enc`` and ``_0..`` (reference criterion/__init__.py:22-40)."""
    weight_dict = {"loss_class": 1, "loss_bbox": 5.0, "loss_giou": 2.0}
    base = copy.deepcopy(weight_dict)
    if args.aux_loss:
        aux = {k + "_enc": v for k, v in base.items()}
        for i in range(args.dec_layers - 1):
            aux.update({k + f"_{i}": v for k, v in base.items()})
        weight_dict.update(aux)
    return TwoStageCriterion(num_classes=args.max_text_len, matcher=build_matcher(args),
                             weight_dict=weight_dict)
