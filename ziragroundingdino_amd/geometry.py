"""Host side of the level-geometry kernels (csrc/refpoints.hip): what the reference derives from the padding mask and the level
table alone -- valid ratios (transformer_for_adapter.py:226-233, :260), the encoder's reference points (:482-497) and the
two-stage proposals (utils.py:56-116) -- as one or two launches each instead of ~150 launch-bound ATen kernels per step (a tiny
launch costs the replayed step 3 us: ``scripts/ab_step.py dummy_launches=N``).  Bit-identical to the op chains
(tests/test_geometry_gpu.py).  Device tensors only; the callers keep their op chains for CPU tensors."""
import torch

from . import _lib

_TABLES = {}


def level_tables(shapes, device):
    """(spatial_shapes [L, 2], level_start_index [L]) int64 on ``device`` for a tuple of (H, W), built once per geometry (the
    reference uploads them on every forward; a cached tensor also keeps the forward capturable into a hipGraph)."""
    key = (tuple((int(h), int(w)) for h, w in shapes), str(device))
    got = _TABLES.get(key)
    if got is None:
        sh = torch.as_tensor(key[0], dtype=torch.long, device=device)
        start = torch.cat((sh.new_zeros((1,)), sh.prod(1).cumsum(0)[:-1]))
        if len(_TABLES) > 64:
            _TABLES.clear()
        got = _TABLES[key] = (sh, start)
    return got


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _check(rc, name):
    if rc != 0:
        raise RuntimeError("%s failed: hipError %d" % (name, rc))


def supported(mask_flat, shapes) -> bool:
    return (torch.is_tensor(mask_flat) and mask_flat.is_cuda and mask_flat.dtype == torch.bool and mask_flat.dim() == 2
            and mask_flat.is_contiguous() and 0 < len(shapes) <= 30
            and sum(int(h) * int(w) for h, w in shapes) == mask_flat.shape[1])


def valid_ratios(mask_flat: torch.Tensor, shapes) -> torch.Tensor:
    """[B, L, 2] = (valid_W / W, valid_H / H) per level of ``mask_flat`` [B, S] (True = padded)."""
    B, S = mask_flat.shape
    sh, start = level_tables(shapes, mask_flat.device)
    out = torch.empty((B, len(shapes), 2), device=mask_flat.device, dtype=torch.float32)
    with torch.cuda.device(mask_flat.device):
        _check(_lib.load().zira_level_valid_ratios_f32(mask_flat.data_ptr(), sh.data_ptr(), start.data_ptr(), B, S, len(shapes), 0,
                                                       out.data_ptr(), _stream(mask_flat)), "zira_level_valid_ratios_f32")
    return out


def encoder_reference_points(ratios: torch.Tensor, shapes) -> torch.Tensor:
    """[B, S, L, 2] from valid ratios [B, L, 2] (fp32, contiguous, on the GPU, no gradient)."""
    B, L, _ = ratios.shape
    S = sum(int(h) * int(w) for h, w in shapes)
    sh, start = level_tables(shapes, ratios.device)
    out = torch.empty((B, S, L, 2), device=ratios.device, dtype=torch.float32)
    with torch.cuda.device(ratios.device):
        _check(_lib.load().zira_encoder_ref_points_f32(ratios.data_ptr(), sh.data_ptr(), start.data_ptr(), B, S, L, out.data_ptr(),
                                                       _stream(ratios)), "zira_encoder_ref_points_f32")
    return out


def encoder_proposals(mask_flat: torch.Tensor, shapes):
    """(proposals [B, S, 4] un-sigmoided with +inf where dropped, drop [B, S] bool) for ``learnedwh`` = None."""
    B, S = mask_flat.shape
    L = len(shapes)
    sh, start = level_tables(shapes, mask_flat.device)
    odds = torch.empty((B, S, 4), device=mask_flat.device, dtype=torch.float32)
    drop = torch.empty((B, S), device=mask_flat.device, dtype=torch.bool)
    scratch = torch.empty((B, L, 2), device=mask_flat.device, dtype=torch.float32)
    with torch.cuda.device(mask_flat.device):
        _check(_lib.load().zira_encoder_proposals_f32(mask_flat.data_ptr(), sh.data_ptr(), start.data_ptr(), B, S, L,
                                                      scratch.data_ptr(), odds.data_ptr(), drop.data_ptr(), _stream(mask_flat)),
               "zira_encoder_proposals_f32")
    return torch.log_(odds), drop
