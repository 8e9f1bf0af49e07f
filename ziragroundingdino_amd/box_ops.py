"""Box helpers on the loss path (reference groundingdino/util/box_ops.py:9-66)."""
import torch


def box_cxcywh_to_xyxy(x):
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def box_xyxy_to_cxcywh(x):
    x0, y0, x1, y1 = x.unbind(-1)
    return torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, x1 - x0, y1 - y0], dim=-1)


def box_area(boxes):
    """torchvision.ops.boxes.box_area (the reference's only use of torchvision on this path)."""
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def box_iou(boxes1, boxes2):
    """Pairwise IoU [N, M] and union, with the reference's +1e-6 in the denominator."""
    area1, area2 = box_area(boxes1), box_area(boxes2)
    lt = torch.max(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.min(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    union = area1[:, None] + area2 - inter
    return inter / (union + 1e-6), union


def generalized_box_iou(boxes1, boxes2, check=True):
    """Pairwise GIoU [N, M] of xyxy boxes (reference box_ops.py:39-66).  The two assertions are
    host syncs; ``check=False`` is for callers that fold the check into a copy they make anyway."""
    if check:
        assert (boxes1[:, 2:] >= boxes1[:, :2]).all(), "boxes1 not in xyxy order"
        assert (boxes2[:, 2:] >= boxes2[:, :2]).all(), "boxes2 not in xyxy order"
    iou, union = box_iou(boxes1, boxes2)
    lt = torch.min(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.max(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = (rb - lt).clamp(min=0)
    area = wh[:, :, 0] * wh[:, :, 1]
    return iou - (area - union) / (area + 1e-6)


def generalized_box_iou_aligned(boxes1, boxes2):
    """GIoU of the aligned pairs (boxes1[i], boxes2[i]) -- the diagonal of
    ``generalized_box_iou`` (what the reference's ``loss_boxes`` extracts with ``torch.diag``,
    criterion.py:176-181) without forming the N x N matrix.  Same arithmetic per pair; callers
    check the xyxy order of the boxes themselves."""
    area1 = (boxes1[:, 2] - boxes1[:, 0]) * (boxes1[:, 3] - boxes1[:, 1])
    area2 = (boxes2[:, 2] - boxes2[:, 0]) * (boxes2[:, 3] - boxes2[:, 1])
    wh = (torch.min(boxes1[:, 2:], boxes2[:, 2:]) - torch.max(boxes1[:, :2], boxes2[:, :2])).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    union = area1 + area2 - inter
    iou = inter / (union + 1e-6)
    wh = (torch.max(boxes1[:, 2:], boxes2[:, 2:]) - torch.min(boxes1[:, :2], boxes2[:, :2])).clamp(min=0)
    area = wh[:, 0] * wh[:, 1]
    return iou - (area - union) / (area + 1e-6)
