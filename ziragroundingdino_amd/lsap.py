"""Batched linear sum assignment on the device (C ABI ``zira_lsap_f32``, csrc/lsap.hip).

Replaces the per-image ``scipy.optimize.linear_sum_assignment`` calls of the reference's matcher
(groundingdino/models/GroundingDINO/matcher/matcher.py:105-151) for all prediction sets of a step:
same assignments for the same float32 costs, ties included, and nothing leaves the device -- the
host only needs the number of targets per image, which it knows from the shapes.
"""
from typing import Sequence, Tuple

import torch

from . import _lib

_meta_cache = {}
_state = {}


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def match_layout(sizes: Sequence[int], num_queries: int, device) -> Tuple[torch.Tensor, int, int, int]:
    """(meta int32 [2 * (B + 1)] on the device, Ttot, Tmax, Mtot) for images with ``sizes`` targets."""
    key = (tuple(sizes), num_queries, str(device))
    hit = _meta_cache.get(key)
    if hit is None:
        toff, moff = [0], [0]
        for n in sizes:
            toff.append(toff[-1] + n)
            moff.append(moff[-1] + min(n, num_queries))
        meta = torch.tensor(toff + moff, dtype=torch.int32).to(device)
        if len(_meta_cache) > 256:
            _meta_cache.clear()
        hit = _meta_cache[key] = (meta, toff[-1], max(sizes) if sizes else 0, moff[-1])
    return hit


def _dev_key(device) -> str:
    """State key of a device: always the indexed form ("cuda" -> "cuda:<current device>")."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return str(device)


def infeasible(device, reset: bool = False) -> bool:
    """True if any assignment on ``device`` since start-up (or the last ``reset``) met an infeasible
    (inf / NaN) cost matrix -- scipy raises ValueError there (status bit 0; bit 1 belongs to ``bad_boxes``).
    Reading the flag synchronises; meant for checkpoints, the end of a task, tests and debugging."""
    st = _state.get(_dev_key(device))
    hit = bool(st is not None and int(st["status"].item()) & 1)
    if reset and st is not None:
        st["status"].bitwise_and_(~1)
    return hit


def linear_sum_assignment_batched(cost: torch.Tensor, sizes: Sequence[int], global_targets: bool = False):
    """``cost`` [S, B, Q, Ttot] float32 on the GPU, image b's targets in columns
    ``sum(sizes[:b]) .. sum(sizes[:b + 1])``.  Returns (q_idx, t_idx) int64 [S, Mtot]: scipy's
    (row_ind, col_ind) of image b in columns ``moff[b] .. moff[b + 1]`` (moff = cumulative min(Q, n_b));
    ``global_targets`` adds the image's column offset to col_ind."""
    assert cost.is_cuda and cost.dtype == torch.float32 and cost.dim() == 4, "expected a float32 [S, B, Q, T] GPU tensor"
    cost = cost.contiguous()
    S, B, Q, T = cost.shape
    assert len(sizes) == B and sum(sizes) == T, "sizes must cover the target columns"
    lib = _lib.load()
    dev = cost.device
    meta, Ttot, Tmax, Mtot = match_layout(sizes, Q, dev)
    q_idx = torch.empty((S, Mtot), dtype=torch.int64, device=dev)
    t_idx = torch.empty((S, Mtot), dtype=torch.int64, device=dev)
    if Mtot == 0:
        return q_idx, t_idx
    st = _state.get(_dev_key(dev))
    need = int(lib.zira_lsap_workspace_bytes(S, B, Q, Tmax))
    if st is None or st["ws"].numel() < need:
        st = _state[_dev_key(dev)] = {"status": st["status"] if st else torch.zeros(1, dtype=torch.int32, device=dev),
                                 "ws": torch.empty(max(need, 16), dtype=torch.uint8, device=dev)}
    with torch.cuda.device(dev):
        rc = lib.zira_lsap_f32(cost.data_ptr(), S, B, Q, Ttot, Tmax, meta.data_ptr(), q_idx.data_ptr(),
                               t_idx.data_ptr(), Mtot, int(global_targets), st["status"].data_ptr(),
                               st["ws"].data_ptr(), st["ws"].numel(), _stream(dev))
    if rc != 0:
        raise RuntimeError("zira_lsap_f32 failed with code %d" % rc)
    return q_idx, t_idx


def _status(dev):
    st = _state.get(_dev_key(dev))
    if st is None:
        st = _state[_dev_key(dev)] = {"status": torch.zeros(1, dtype=torch.int32, device=dev),
                                 "ws": torch.empty(16, dtype=torch.uint8, device=dev)}
    return st["status"]


def bad_boxes(device, reset: bool = False) -> bool:
    """True if ``matching_cost`` met a box with x1 < x0 or y1 < y0 (the reference asserts there; synchronises)."""
    st = _state.get(_dev_key(device))
    hit = bool(st is not None and int(st["status"].item()) & 2)
    if reset and st is not None:
        st["status"].bitwise_and_(~2)
    return hit


def matching_cost(logits, boxes, tgt_ids, tgt_boxes, w_class=1.0, w_bbox=1.0, w_giou=1.0, alpha=0.25, gamma=2.0):
    """Fused focal + L1 + GIoU matching cost (C ABI ``zira_match_cost_f32``; reference matcher.py:105-141):
    logits [N, C], boxes [N, 4], tgt_ids [T], tgt_boxes [T, 4] -> [N, T] float32."""
    assert logits.is_cuda and logits.dtype == torch.float32 and boxes.dtype == torch.float32
    lib = _lib.load()
    logits, boxes = logits.contiguous(), boxes.contiguous()
    tgt_ids, tgt_boxes = tgt_ids.contiguous().to(torch.int64), tgt_boxes.contiguous().float()
    N, C = logits.shape
    T = tgt_ids.numel()
    cost = torch.empty((N, T), dtype=torch.float32, device=logits.device)
    with torch.cuda.device(logits.device):
        rc = lib.zira_match_cost_f32(logits.data_ptr(), boxes.data_ptr(), tgt_ids.data_ptr(), tgt_boxes.data_ptr(), N, C, T,
                                     w_class, w_bbox, w_giou, alpha, gamma, cost.data_ptr(),
                                     _status(logits.device).data_ptr(), _stream(logits.device))
    if rc != 0:
        raise RuntimeError("zira_match_cost_f32 failed with code %d" % rc)
    return cost
