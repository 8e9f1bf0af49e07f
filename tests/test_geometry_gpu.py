"""Level geometry from the padding mask (csrc/refpoints.hip through geometry.py) against the op chains of the reference that
it stands for -- get_valid_ratio (transformer_for_adapter.py:226-233), get_reference_points (:482-497),
gen_encoder_output_proposals (utils.py:56-116) -- which the package keeps for CPU tensors: bit-identical."""
import pytest
import torch

from ziragroundingdino_amd import geometry, transformer as zt, utils

pytestmark = pytest.mark.gpu

SHAPES = [[(100, 167), (50, 84), (25, 42), (13, 21)], [(7, 9), (4, 5)], [(1, 1), (3, 1), (1, 5)], [(64, 64)]]


def _masks(shapes, B, kind, gen):
    out = []
    for b in range(B):
        per = []
        for (H, W) in shapes:
            m = torch.zeros(H, W, dtype=torch.bool)
            if kind == "padded":
                h = int(torch.randint(max(1, H // 2), H + 1, (1,), generator=gen))
                w = int(torch.randint(max(1, W // 2), W + 1, (1,), generator=gen))
                m[h:, :] = True
                m[:, w:] = True
            elif kind == "ragged":          # not a rectangle: the chains only look at the first row / column
                m = torch.rand(H, W, generator=gen) < 0.3
                m[0, 0] = False
            per.append(m.flatten())
        out.append(torch.cat(per))
    return torch.stack(out).cuda()


@pytest.mark.parametrize("shapes", SHAPES)
@pytest.mark.parametrize("kind", ["none", "padded", "ragged"])
def test_geometry_matches_the_op_chains_bit_for_bit(shapes, kind):
    gen = torch.Generator().manual_seed(len(shapes) * 7 + len(kind))
    B = 3
    mask = _masks(shapes, B, kind, gen)
    assert geometry.supported(mask, shapes)
    S = mask.shape[1]
    memory = torch.randn(B, S, 8, generator=gen).cuda()
    per_level, cur = [], 0
    for (H, W) in shapes:
        per_level.append(mask[:, cur:cur + H * W].view(B, H, W))
        cur += H * W
    old = zt.Switches.native_geometry
    try:
        zt.Switches.native_geometry = False
        vr_ref = torch.stack([zt.Transformer.get_valid_ratio(m) for m in per_level], 1)
        rp_ref = zt.TransformerEncoder.get_reference_points(shapes, vr_ref, mask.device)
        mem_ref, prop_ref = utils.gen_encoder_output_proposals(memory, mask, shapes)
        zt.Switches.native_geometry = True
        vr = geometry.valid_ratios(mask, shapes)
        rp = zt.TransformerEncoder.get_reference_points(shapes, vr, mask.device)
        mem, prop = utils.gen_encoder_output_proposals(memory, mask, shapes)
    finally:
        zt.Switches.native_geometry = old
    assert torch.equal(vr, vr_ref)
    assert rp.shape == rp_ref.shape and torch.equal(rp.nan_to_num(nan=-7.0), rp_ref.nan_to_num(nan=-7.0))
    assert prop.shape == prop_ref.shape and torch.equal(prop, prop_ref)
    assert torch.equal(mem, mem_ref)


def test_proposal_memory_gradient_is_masked_like_the_reference():
    shapes = SHAPES[1]
    gen = torch.Generator().manual_seed(3)
    mask = _masks(shapes, 2, "padded", gen)
    memory = torch.randn(2, mask.shape[1], 4, generator=gen).cuda()
    grads = []
    for native in (False, True):
        old = zt.Switches.native_geometry
        zt.Switches.native_geometry = native
        try:
            m = memory.clone().requires_grad_(True)
            out, _ = utils.gen_encoder_output_proposals(m, mask, shapes)
            out.square().sum().backward()
            grads.append(m.grad)
        finally:
            zt.Switches.native_geometry = old
    assert torch.equal(grads[0], grads[1])


@pytest.mark.parametrize("hw", [(100, 167), (13, 21), (1, 1), (7, 300)])
@pytest.mark.parametrize("normalize", [True, False])
def test_sine_position_encoding_is_bit_identical_to_the_op_chain(hw, normalize):
    """PositionEmbeddingSineHW (position_encoding.py:78-134) in one launch per level against its ATen op chain, which the
    package keeps for CPU masks: rectangular padding, ragged masks and fully padded rows / columns."""
    from ziragroundingdino_amd import backbone
    from ziragroundingdino_amd.utils import NestedTensor
    H, W = hw
    gen = torch.Generator().manual_seed(H * 31 + W)
    masks = torch.zeros(3, H, W, dtype=torch.bool)
    masks[1, H - H // 3:, :] = True
    masks[1, :, W - W // 4:] = True
    masks[2] = torch.rand(H, W, generator=gen) < 0.3
    masks = masks.cuda()
    pe = backbone.PositionEmbeddingSineHW(128, temperatureH=20, temperatureW=20, normalize=normalize)
    x = NestedTensor(torch.zeros(3, 4, H, W, device="cuda"), masks)
    pe.native = False
    ref = pe(x)
    pe.native = True
    got = pe(x)
    assert got.shape == ref.shape
    assert torch.equal(got, ref)
