"""The focal / L1 / GIoU losses of all prediction sets as one native node (csrc/criterion.hip through
criterion._StackedLosses) against the op chain it replaces (``native_losses = False``: the ATen form of the reference's
SetCriterion.loss_labels / loss_boxes, criterion/criterion.py:104-181, itself pinned to the reference's golden losses by
tests/test_modules_golden.py): losses and both gradients, at the training step's size and with matched boxes that sit exactly
on their targets (the kinks of min / max / abs, where autograd's conventions decide)."""
from types import SimpleNamespace

import pytest
import torch

from ziragroundingdino_amd import criterion

pytestmark = pytest.mark.gpu


def _case(S, B, Q, C, sizes, seed, exact_pairs=False):
    g = torch.Generator().manual_seed(seed)
    logits = (torch.randn(S, B, Q, C, generator=g) * 2).cuda()
    logits[..., C // 2:] = -100.0          # the fill of recover_to_cls_logits for categories an image does not have
    cxcy = torch.rand(S, B, Q, 2, generator=g) * 0.6 + 0.2
    wh = torch.rand(S, B, Q, 2, generator=g) * 0.3 + 0.05
    boxes = torch.cat([cxcy, wh], -1).cuda()
    targets = []
    for n in sizes:
        c = torch.rand(n, 2, generator=g) * 0.5 + 0.25
        w = torch.rand(n, 2, generator=g) * 0.3 + 0.1
        targets.append({"labels": torch.randint(0, C // 2, (n,), generator=g).cuda(), "boxes": torch.cat([c, w], -1).cuda()})
    if exact_pairs:                        # a few predictions ARE a target (or share one of its edges)
        for b, t in enumerate(targets):
            for k in range(min(3, len(t["boxes"]))):
                boxes[:, b, k] = t["boxes"][k]
            if len(t["boxes"]) > 3:
                boxes[:, b, 3, :2] = t["boxes"][3, :2]
    return logits, boxes, targets


@pytest.mark.parametrize("shape", [(7, 2, 900, 256, (6, 4), False), (7, 2, 40, 32, (5, 3), False), (3, 3, 70, 64, (9, 0, 2), False),
                                   (7, 2, 40, 32, (5, 3), True), (2, 1, 5, 8, (9,), False),
                                   (3, 2, 300, 16, (90, 75), False)])     # 165 pairs per set: more than one wave of candidates per row
def test_native_losses_match_the_op_chain(shape):
    S, B, Q, C, sizes, exact = shape
    logits, boxes, targets = _case(S, B, Q, C, sizes, seed=S * 100 + Q, exact_pairs=exact)
    crit = criterion.build_criterion(SimpleNamespace(aux_loss=True, dec_layers=S - 1, max_text_len=C)).cuda()
    suffixes = ["_%d" % i for i in range(S - 2)] + ["", "_enc"]
    res = {}
    for native in (False, True):
        crit.native_losses = native
        lg, bx = logits.clone().requires_grad_(True), boxes.clone().requires_grad_(True)
        out = {"stacked": (lg, bx, suffixes)}
        losses = crit(out, targets)
        w = torch.linspace(0.5, 2.0, S, device="cuda")
        total = sum((vec * w).sum() * (i + 1) for i, (name, (vec, _)) in enumerate(sorted(losses.stacked.items())))
        gl, gb = torch.autograd.grad(total, [lg, bx])
        res[native] = ({k: v.detach() for k, v in losses.items()}, gl, gb)
    ref, got = res[False], res[True]
    assert set(ref[0]) == set(got[0]) and len(got[0]) == 3 * S
    for k in ref[0]:
        assert torch.allclose(got[0][k], ref[0][k], rtol=2e-6, atol=1e-7), (k, got[0][k].item(), ref[0][k].item())
    # gradients: 1e-5 of the largest entry (the chain sums its contributions in another order), exact zeros where it has them
    for name, a, b in (("logits", got[1], ref[1]), ("boxes", got[2], ref[2])):
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() <= 1e-5 * scale + 1e-12, (name, (a - b).abs().max().item(), scale)
        if name == "boxes":
            assert torch.equal(a == 0, b == 0)
