"""Hungarian matching of predictions to ground truth (reference
groundingdino/models/GroundingDINO/matcher/matcher.py:27-151; ``build_matcher`` uses the
defaults, i.e. weights 1/1/1, matcher/__init__.py:20-21).

On GPU tensors the criterion uses ``forward_stacked_device``: the assignments of all prediction sets
are solved on the device (csrc/lsap.hip, scipy's algorithm and tie-breaking) and nothing is copied to
the host.  The host paths (``forward`` / ``forward_many`` / ``forward_stacked``) keep scipy's
``linear_sum_assignment`` on one device->host copy, as in the reference; the index results are
bit-identical whenever the cost matrix is.
"""
import torch
import torch.nn as nn
from scipy.optimize import linear_sum_assignment

from .box_ops import box_cxcywh_to_xyxy, generalized_box_iou


class HungarianMatcher(nn.Module):
    def __init__(self, cost_class: float = 1, cost_bbox: float = 1, cost_giou: float = 1,
                 cost_class_type: str = "focal_loss_cost", alpha: float = 0.25, gamma: float = 2.0):
        super().__init__()
        assert cost_class != 0 or cost_bbox != 0 or cost_giou != 0, "all costs cant be 0"
        assert cost_class_type in {"ce_cost", "focal_loss_cost"}
        self.cost_class = cost_class
        self.cost_bbox = cost_bbox
        self.cost_giou = cost_giou
        self.cost_class_type = cost_class_type
        self.alpha = alpha
        self.gamma = gamma

    @torch.no_grad()
    def cost_matrix(self, outputs, targets, check=True):
        """[bs, num_queries, total_targets] matching cost (device tensor).  ``check=False`` returns
        (cost, ok) with the xyxy-order check as a device scalar instead of asserting it."""
        bs, num_queries = outputs["pred_logits"].shape[:2]
        logits = outputs["pred_logits"].flatten(0, 1)
        out_bbox = outputs["pred_boxes"].flatten(0, 1)
        tgt_ids = torch.cat([v["labels"] for v in targets])
        tgt_bbox = torch.cat([v["boxes"] for v in targets])
        if self.cost_class_type == "ce_cost":
            cost_class = -logits.softmax(-1)[:, tgt_ids]
        else:
            p = logits.sigmoid()
            neg = (1 - self.alpha) * (p ** self.gamma) * (-(1 - p + 1e-8).log())
            pos = self.alpha * ((1 - p) ** self.gamma) * (-(p + 1e-8).log())
            cost_class = pos[:, tgt_ids] - neg[:, tgt_ids]
        cost_bbox = torch.cdist(out_bbox, tgt_bbox, p=1)
        b1, b2 = box_cxcywh_to_xyxy(out_bbox), box_cxcywh_to_xyxy(tgt_bbox)
        if check:
            cost_giou = -generalized_box_iou(b1, b2)
        else:
            ok = (b1[:, 2:] >= b1[:, :2]).all() & (b2[:, 2:] >= b2[:, :2]).all()
            cost_giou = -generalized_box_iou(b1, b2, check=False)
        C = self.cost_bbox * cost_bbox + self.cost_class * cost_class + self.cost_giou * cost_giou
        C = C.view(bs, num_queries, -1)
        return C if check else (C, ok)

    @torch.no_grad()
    def forward_stacked(self, logits, boxes, targets):
        """``forward`` for S prediction sets given as stacked tensors ``logits [S, B, Q, C]`` /
        ``boxes [S, B, Q, 4]``: one cost computation over all S*B*Q predictions, one device->host
        copy, S*B assignments.  The xyxy-order assertion of ``generalized_box_iou`` travels with
        that copy instead of being its own host sync."""
        S, B, Q = logits.shape[:3]
        flat = {"pred_logits": logits.reshape(1, S * B * Q, -1), "pred_boxes": boxes.reshape(1, S * B * Q, 4)}
        C, ok = self.cost_matrix(flat, targets, check=False)
        host = torch.cat([C.reshape(-1), ok.to(C.dtype).reshape(1)]).cpu()
        assert bool(host[-1]), "boxes not in xyxy order"
        C = host[:-1].view(S, B, Q, -1)
        sizes = [len(v["boxes"]) for v in targets]
        results = []
        for s in range(S):
            indices = [linear_sum_assignment(c[i]) for i, c in enumerate(C[s].split(sizes, -1))]
            results.append([(torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64))
                            for i, j in indices])
        return results

    @torch.no_grad()
    def forward_stacked_device(self, logits, boxes, targets):
        """``forward_stacked`` without leaving the GPU: the S*B assignments are solved by the device-side
        solver (csrc/lsap.hip: scipy's algorithm and tie-breaking, one wavefront per problem), so the
        criterion has no device->host copy and no host synchronisation left.  Returns
        (q_idx, t_idx) int64 ``[S, M]`` device tensors, M = sum over images of min(Q, targets): set s
        matches query ``q_idx[s, k]`` of image ``image_of[k]`` with target ``t_idx[s, k]`` of the
        CONCATENATED targets.  The xyxy-order assertion of ``generalized_box_iou`` (a host sync of its own
        in the reference) is kept as a device flag: ``check()`` raises if it ever failed."""
        from .lsap import linear_sum_assignment_batched, matching_cost

        S, B, Q = logits.shape[:3]
        sizes = [len(v["boxes"]) for v in targets]
        self._device = logits.device
        if self.cost_class_type == "focal_loss_cost" and logits.dtype == torch.float32 and sum(sizes):
            # one kernel for the whole cost (csrc/lsap.hip), rounded like the chain of ~30 PyTorch kernels below
            C = matching_cost(logits.detach().reshape(S * B * Q, -1), boxes.detach().reshape(S * B * Q, 4).float(),
                              torch.cat([v["labels"] for v in targets]), torch.cat([v["boxes"] for v in targets]),
                              self.cost_class, self.cost_bbox, self.cost_giou, self.alpha, self.gamma)
        else:
            flat = {"pred_logits": logits.reshape(1, S * B * Q, -1), "pred_boxes": boxes.reshape(1, S * B * Q, 4)}
            C, ok = self.cost_matrix(flat, targets, check=False)
            bad = getattr(self, "_bad_boxes", None)
            self._bad_boxes = ~ok if bad is None or bad.device != ok.device else bad | ~ok
        return linear_sum_assignment_batched(C.view(S, B, Q, -1), sizes, global_targets=True)

    def check(self):
        """Raise if a prediction / target box of any device-side matching so far was not in xyxy order, or a
        cost matrix was infeasible (the reference asserts / scipy raises on the spot; this reads two flags
        back and therefore synchronises -- call it at the end of a task, not in the step)."""
        from .lsap import bad_boxes, infeasible

        bad = getattr(self, "_bad_boxes", None)
        assert bad is None or not bool(bad), "boxes not in xyxy order"
        dev = getattr(self, "_device", None)
        if dev is not None:
            assert not bad_boxes(dev), "boxes not in xyxy order"
            if infeasible(dev):
                raise ValueError("cost matrix is infeasible")

    @torch.no_grad()
    def forward(self, outputs, targets):
        """-> list (per image) of (query_idx, target_idx) int64 CPU tensors."""
        C = self.cost_matrix(outputs, targets).cpu()
        sizes = [len(v["boxes"]) for v in targets]
        indices = [linear_sum_assignment(c[i]) for i, c in enumerate(C.split(sizes, -1))]
        return [(torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64))
                for i, j in indices]


    @torch.no_grad()
    def forward_many(self, outputs_list, targets):
        """Match several prediction sets (final layer, auxiliary layers, encoder proposals)
        against the same targets with ONE device->host copy: all cost matrices are built on the
        GPU first, the linear sum assignments then run on the host back to back.  Same
        assignments as calling ``forward`` once per set (the reference does that: 7 syncs)."""
        costs = torch.stack([self.cost_matrix(o, targets) for o in outputs_list]).cpu()
        sizes = [len(v["boxes"]) for v in targets]
        results = []
        for C in costs:
            indices = [linear_sum_assignment(c[i]) for i, c in enumerate(C.split(sizes, -1))]
            results.append([(torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64))
                            for i, j in indices])
        return results


def build_matcher(args=None):
    return HungarianMatcher()
