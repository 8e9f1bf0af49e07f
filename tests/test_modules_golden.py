"""Modules around the MSDA op against golden vectors produced by the reference itself
(tests/golden/gen_modules_golden.py -> mod_*.pt).

Every case runs twice: on the CPU (host logic; the native MSDA entry points are replaced by the
CPU oracle through a test-only monkeypatch of ``ziragroundingdino_amd._C`` -- the product has
no such path) and, marked ``gpu``, on the MI355X with the real HIP kernels.
Tolerances: fp32 1e-4 relative to the tensor scale for module outputs / gradients (north_star:
1e-3); index and mask results (text masks, matcher assignments, two-stage top-k) bit-exact.
"""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
from seeded import fill_by_name_  # noqa: E402

import ziragroundingdino_amd as z  # noqa: E402
from ziragroundingdino_amd import _C, criterion, matcher, rsb, text_masks, transformer, utils  # noqa: E402

DEVICES = ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)]
TOL = 1e-4


def load(name):
    return torch.load(os.path.join(GOLDEN, "mod_%s.pt" % name), weights_only=False)


def to(obj, dev):
    if torch.is_tensor(obj):
        return obj.to(dev)
    if isinstance(obj, dict):
        return {k: to(v, dev) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(to(v, dev) for v in obj)
    return obj


def close(got, want, tol=TOL, what=""):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    fin = torch.isfinite(want)
    assert torch.equal(torch.isfinite(got), fin), what + ": inf/nan pattern differs"
    assert torch.equal(got[~fin], want[~fin]), what
    scale = max(1.0, float(want[fin].abs().max())) if fin.any() else 1.0
    err = float((got[fin] - want[fin]).abs().max()) / scale if fin.any() else 0.0
    assert err <= tol, "%s: max err %.3e (scaled) > %.1e" % (what, err, tol)


@pytest.fixture
def msda_backend(request, monkeypatch, oracle):
    """device string; on 'cpu' the two native entry points are served by the oracle."""
    dev = request.param
    if dev == "cpu":
        def fwd(value, shapes, start, loc, attn, step):
            out = oracle.msda_forward(value.detach().numpy(), shapes.numpy(), start.numpy(),
                                      loc.detach().numpy(), attn.detach().numpy())
            return torch.from_numpy(out)

        def bwd(value, shapes, start, loc, attn, go, step):
            g = oracle.msda_backward(go.detach().numpy(), value.detach().numpy(), shapes.numpy(),
                                     start.numpy(), loc.detach().numpy(), attn.detach().numpy())
            return [torch.from_numpy(x) for x in g]

        monkeypatch.setattr(_C, "ms_deform_attn_forward", fwd)
        monkeypatch.setattr(_C, "ms_deform_attn_backward", bwd)
    return dev


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("msda_backend", DEVICES, indirect=True)
@pytest.mark.parametrize("refdim", [2, 4])
def test_msda_module(msda_backend, refdim):
    dev = msda_backend
    g = to(load("msda_module_ref%d" % refdim), dev)
    mod = z.MultiScaleDeformableAttention(embed_dim=64, num_heads=4, num_levels=3, num_points=2,
                                          batch_first=True)
    # deterministic part of init_weights(): offsets grid, zero attention weights/bias
    for k in ("sampling_offsets.weight", "sampling_offsets.bias", "attention_weights.weight",
              "attention_weights.bias", "value_proj.bias", "output_proj.bias"):
        close(mod.state_dict()[k], g["init_state"][k], 1e-6, "init " + k)
    mod.load_state_dict(g["state"])
    mod.to(dev)
    q = g["query"].clone().requires_grad_(True)
    v = g["value"].clone().requires_grad_(True)
    out = mod(query=q, value=v, query_pos=g["query_pos"], key_padding_mask=g["key_padding_mask"],
              reference_points=g["reference_points"], spatial_shapes=g["spatial_shapes"],
              level_start_index=g["level_start_index"])
    close(out, g["out"], TOL, "out")
    params = dict(mod.named_parameters())
    grads = torch.autograd.grad(out, [q, v] + list(params.values()), g["grad_out"])
    close(grads[0], g["grad_query"], TOL, "grad_query")
    close(grads[1], g["grad_value"], TOL, "grad_value")
    for (k, _), gr in zip(params.items(), grads[2:]):
        close(gr, g["grad_params"][k], TOL, "grad " + k)


@pytest.mark.parametrize("dev", DEVICES)
def test_rep_zero_linear(dev):
    g = to(load("rep_zero_linear"), dev)
    lin = rsb.RepZeroLinear(24, 16)
    # constructor contract: weight 1e-8, scaling 0.1, zero twin; branch bias keeps nn.Linear init
    assert torch.all(lin.weight == 1e-8) and float(lin.scaling) == pytest.approx(0.1)
    assert not lin.freeze_linear.weight.any() and not lin.freeze_linear.bias.any()
    assert set(lin.state_dict()) == set(g["init_state"])
    lin.load_state_dict(g["state"])
    lin.to(dev).train()
    x = g["x"].clone().requires_grad_(True)
    out, zl = lin(x)
    close(out, g["out"], TOL, "out")
    close(zl, g["zl"], TOL, "zero-interference loss")
    assert zl.dim() == 0
    params = dict(lin.named_parameters())
    grads = torch.autograd.grad((out * g["grad_out"]).sum() + g["zl_weight"] * zl, [x] + list(params.values()))
    close(grads[0], g["grad_x"], TOL, "grad_x")
    for k, gr in zip(params, grads[1:]):
        close(gr, g["grad_params"][k], TOL, "grad " + k)
    lin.eval()
    out_e, zl_e = lin(x)
    close(out_e, g["out_eval"], TOL, "eval out")
    assert zl_e.shape == (1,) and float(zl_e) == 0.0
    lin.__rep__()
    for k, v in lin.state_dict().items():
        close(v, g["state_after_rep"][k], 1e-6, "after __rep__ " + k)


@pytest.mark.parametrize("dev", DEVICES)
@pytest.mark.parametrize("name", ["1x1", "3x3s2"])
def test_rep_zero_conv(dev, name):
    g = to(load("rep_zero_conv_" + name), dev)
    conv = rsb.RepZeroConv2d(12, 16, **g["kwargs"])
    assert torch.all(conv.weight == 1e-8) and torch.all(conv.bias == 1e-8)
    assert float(conv.scaling) == pytest.approx(0.1)
    assert set(conv.state_dict()) == set(g["init_state"])
    conv.load_state_dict(g["state"])
    conv.to(dev).train()
    x = g["x"].clone().requires_grad_(True)
    out, zl = conv(x)
    close(out, g["out"], TOL, "out")
    close(zl, g["zl"], TOL, "zero-interference loss")
    params = dict(conv.named_parameters())
    grads = torch.autograd.grad((out * g["grad_out"]).sum() + g["zl_weight"] * zl, [x] + list(params.values()))
    close(grads[0], g["grad_x"], TOL, "grad_x")
    for k, gr in zip(params, grads[1:]):
        close(gr, g["grad_params"][k], TOL, "grad " + k)
    conv.eval()
    out_e, zl_e = conv(x)
    close(out_e, g["out_eval"], TOL, "eval out")
    assert float(zl_e) == 0.0
    conv.__rep__()
    for k, v in conv.state_dict().items():
        close(v, g["state_after_rep"][k], 1e-6, "after __rep__ " + k)


@pytest.mark.parametrize("dev", DEVICES)
def test_text_masks_bit_exact(dev):
    g = to(load("text_masks"), dev)
    for suffix in ("", "2", "3"):
        am, pid, c2t = text_masks.generate_masks_with_special_tokens_and_transfer_map(
            {"input_ids": g["input_ids" + suffix]}, g["special"], None)
        assert am.dtype == torch.bool and pid.dtype == torch.long
        assert torch.equal(am, g["attention_mask" + suffix])
        assert torch.equal(pid, g["position_ids" + suffix])
        assert len(c2t) == len(g["cate_to_token" + suffix])
        for a, b in zip(c2t, g["cate_to_token" + suffix]):
            assert torch.equal(a, b)


@pytest.mark.parametrize("dev", DEVICES)
def test_contrastive_embed_and_category_logits(dev):
    g = to(load("contrastive_logits"), dev)
    hs = g["hs"].clone().requires_grad_(True)
    text = g["text"].clone().requires_grad_(True)
    ce = utils.ContrastiveEmbed(max_text_len=g["max_text_len"])
    logits = ce(hs, {"encoded_text": text, "text_token_mask": g["text_token_mask"]})
    close(logits, g["token_logits"], TOL, "token logits")
    cls = utils.recover_to_cls_logits(logits, g["cate_to_token"], for_fill=-100.0)
    close(cls, g["cls_logits"], TOL, "category logits")
    ghs, gtext = torch.autograd.grad(cls, [hs, text], g["grad_out"])
    close(ghs, g["grad_hs"], TOL, "grad hs")
    close(gtext, g["grad_text"], TOL, "grad text")


@pytest.mark.parametrize("dev", DEVICES)
def test_utils_misc(dev):
    g = to(load("utils_misc"), dev)
    close(utils.gen_sineembed_for_position(g["pos4"]), g["sine4"], 1e-5, "sine4")
    close(utils.gen_sineembed_for_position(g["pos4"][..., :2]), g["sine2"], 1e-5, "sine2")
    close(utils.get_sine_pos_embed(g["pos1"], num_pos_feats=16, exchange_xy=False), g["sine1"], 1e-5, "sine1")
    om, op = utils.gen_encoder_output_proposals(g["memory"], g["padding_mask"], g["shapes"])
    close(om, g["out_memory"], 1e-6, "proposal memory")
    close(op, g["out_proposals"], 1e-5, "proposals")
    om2, op2 = utils.gen_encoder_output_proposals(g["memory"], g["padding_mask"],
                                                  torch.tensor(g["shapes"], device=dev))
    assert torch.equal(op, op2) and torch.equal(om, om2)
    close(utils.inverse_sigmoid(g["inv_sig_in"]), g["inv_sig_out"], 1e-6, "inverse_sigmoid")


@pytest.mark.parametrize("dev", DEVICES)
def test_criterion_and_matcher(dev):
    from types import SimpleNamespace

    g = to(load("criterion"), dev)
    crit = criterion.build_criterion(SimpleNamespace(**g["args"]))
    assert dict(crit.weight_dict) == g["weight_dict"]
    out = g["outputs"]
    out["pred_logits"] = out["pred_logits"].clone().requires_grad_(True)
    out["pred_boxes"] = out["pred_boxes"].clone().requires_grad_(True)
    losses, idx = crit(out, g["targets"], return_indices=True)
    assert set(losses) == set(g["losses"])
    for k in losses:
        close(losses[k], g["losses"][k], TOL, k)

    def same(a, b):  # matcher assignments: bit-exact
        assert len(a) == len(b)
        for (i1, j1), (i2, j2) in zip(a, b):
            assert torch.equal(i1.cpu(), i2.cpu()) and torch.equal(j1.cpu(), j2.cpu())

    same(idx["indices"], g["indices"]["indices"])
    for a, b in zip(idx["aux_outputs"], g["indices"]["aux_outputs"]):
        same(a, b)
    same(idx["enc_outputs"][0], g["indices"]["enc_outputs"][0])
    total = sum(losses[k] * crit.weight_dict[k] for k in losses)
    close(total, g["total"], TOL, "weighted total")
    gl, gb = torch.autograd.grad(total, [out["pred_logits"], out["pred_boxes"]])
    close(gl, g["grad_pred_logits"], TOL, "grad logits")
    close(gb, g["grad_pred_boxes"], TOL, "grad boxes")

    # the step's path: all sets stacked; on the GPU the assignments are solved on the device (no scipy call)
    aux = out["aux_outputs"]
    logits = torch.stack([a["pred_logits"] for a in aux] + [out["pred_logits"], out["enc_outputs"]["pred_logits"]])
    boxes = torch.stack([a["pred_boxes"] for a in aux] + [out["pred_boxes"], out["enc_outputs"]["pred_boxes"]])
    suffixes = ["_%d" % i for i in range(len(aux))] + ["", "_enc"]
    if dev == "cuda":
        import scipy.optimize as so
        import ziragroundingdino_amd.matcher as zm

        def no_scipy(*a, **k):
            raise AssertionError("linear_sum_assignment called on the GPU path")
        orig, zm.linear_sum_assignment = zm.linear_sum_assignment, no_scipy
    try:
        fast, fidx = crit(dict(out, stacked=(logits, boxes, suffixes)), g["targets"], return_indices=True)
    finally:
        if dev == "cuda":
            zm.linear_sum_assignment = orig
    same(fidx["indices"], g["indices"]["indices"])
    for a, b in zip(fidx["aux_outputs"], g["indices"]["aux_outputs"]):
        same(a, b)
    same(fidx["enc_outputs"][0], g["indices"]["enc_outputs"][0])
    for k in fast:
        close(fast[k], g["losses"][k], TOL, "stacked " + k)


@pytest.mark.parametrize("dev", DEVICES)
def test_bi_attention_block(dev):
    g = to(load("bi_attention_block"), dev)
    blk = transformer.BiAttentionBlock(v_dim=32, l_dim=32, embed_dim=64, num_heads=4, dropout=0.0, drop_path=0.1)
    blk.load_state_dict(g["state"])
    blk.to(dev).eval()
    v = g["v"].clone().requires_grad_(True)
    l = g["l"].clone().requires_grad_(True)
    ov, ol = blk(v, l, attention_mask_v=g["mask_v"], attention_mask_l=g["mask_l"])
    close(ov, g["out_v"], TOL, "out_v")
    close(ol, g["out_l"], TOL, "out_l")
    gv, gl = torch.autograd.grad((ov * g["grad_out_v"]).sum() + (ol * g["grad_out_l"]).sum(), [v, l])
    close(gv, g["grad_v"], TOL, "grad_v")
    close(gl, g["grad_l"], TOL, "grad_l")


@pytest.mark.parametrize("dev", DEVICES)
def test_text_enhancer_layer(dev):
    g = to(load("text_enhancer_layer"), dev)
    lay = transformer.TransformerEncoderLayer(d_model=32, nhead=4, dim_feedforward=48, dropout=0.0)
    lay.load_state_dict(g["state"])
    lay.to(dev).eval()
    src = g["src"].clone().requires_grad_(True)
    out = lay(src, src_mask=~g["may_attend"], src_key_padding_mask=None, pos=g["pos"])
    close(out, g["out"], TOL, "out")
    gs, = torch.autograd.grad(out, [src], g["grad_out"])
    close(gs, g["grad_src"], TOL, "grad_src")


@pytest.mark.parametrize("msda_backend", DEVICES, indirect=True)
def test_tiny_transformer(msda_backend):
    """Encoder (fusion + text enhancer + deformable layer) -> two-stage top-k -> decoder with
    iterative box refinement, weights rebuilt from names (tests/golden/seeded.py)."""
    dev = msda_backend
    g = load("tiny_transformer")
    kw = g["kwargs"]
    d = kw["d_model"]
    tr = transformer.Transformer(**kw)
    bbox = utils.MLP(d, d, 4, 3)
    cls = utils.ContrastiveEmbed(max_text_len=16)
    tr.decoder.bbox_embed = torch.nn.ModuleList([bbox for _ in range(2)])
    tr.decoder.class_embed = torch.nn.ModuleList([cls for _ in range(2)])
    tr.enc_out_bbox_embed = utils.MLP(d, d, 4, 3)
    tr.enc_out_class_embed = cls
    assert [n for n, _ in tr.named_parameters()] == g["param_names"]   # state-dict contract
    fill_by_name_(tr, g["salt"], g["scale"], g["scales"])
    tr.to(dev).eval()
    g = to(g, dev)
    srcs = [s.clone().requires_grad_(True) for s in g["srcs"]]
    text = g["text"].clone().requires_grad_(True)
    text_dict = {"encoded_text": text, "text_token_mask": g["text_token_mask"],
                 "position_ids": g["position_ids"],
                 "text_self_attention_masks": g["text_self_attention_masks"]}
    hs, refs, hs_enc, ref_enc, init_box, _ = tr(srcs, g["masks"], None, g["poss"], None, None, text_dict)
    assert torch.equal(tr.last_topk_proposals, g["topk_proposals"])     # indices: bit-exact
    close(text_dict["encoded_text"], g["memory_text"], TOL, "memory_text")
    for i, (a, b) in enumerate(zip(hs, g["hs"])):
        close(a, b, TOL, "hs[%d]" % i)
    for i, (a, b) in enumerate(zip(refs, g["references"])):
        close(a, b, TOL, "references[%d]" % i)
    close(hs_enc, g["hs_enc"], TOL, "hs_enc")
    close(ref_enc, g["ref_enc"], TOL, "ref_enc")
    close(init_box, g["init_box_proposal"], TOL, "init_box_proposal")
    total = sum((h * go).sum() for h, go in zip(hs, g["grad_hs"])) + (refs[-1] ** 2).sum() + (hs_enc ** 2).sum() * 0.1
    close(total, g["total"], TOL, "total")
    grads = torch.autograd.grad(total, srcs + [text])
    for i in range(3):
        close(grads[i], g["grad_srcs"][i], 2e-4, "grad srcs[%d]" % i)
    close(grads[3], g["grad_text"], 2e-4, "grad text")


@pytest.mark.parametrize("dev", DEVICES)
def test_rsb_zero_init_tiny_gradient_regime(dev):
    """BASELINE configs[4] (1-shot softfreeze): at task start the branch weights are 1e-8 and the
    twin is zero, so activations and gradients of the side branch are ~1e-8..1e-7.  The fused
    epilogue must keep full fp32 relative accuracy there (no flush, no absolute-epsilon shortcuts);
    reference values are the defining expression in float64."""
    torch.manual_seed(3)
    conv = rsb.RepZeroConv2d(48, 64, kernel_size=1).to(dev).train()       # ZiRa init: W=b=1e-8, twin 0
    x = (torch.randn(2, 48, 9, 14) * 3).to(dev)
    out, zl = conv(x)
    go = torch.randn_like(out)
    grads = torch.autograd.grad((out * go).sum() + 0.1 * zl, list(conv.parameters()))
    # float64 restatement on the CPU
    c64 = rsb.RepZeroConv2d(48, 64, kernel_size=1).double().train()
    x64, go64 = x.detach().cpu().double(), go.detach().cpu().double()
    br = c64.scaling * torch.nn.functional.conv2d(x64, c64.weight, c64.bias)
    o64 = br + c64.freeze_conv(x64)
    sl1 = lambda t: torch.where(t.abs() < 1, 0.5 * t * t, t.abs() - 0.5).mean()
    z64 = sl1(br) + sl1(o64)
    g64 = torch.autograd.grad((o64 * go64).sum() + 0.1 * z64, list(c64.parameters()))
    rel = lambda a, b: float((a.detach().cpu().double() - b).abs().max() / b.abs().max().clamp_min(1e-300))
    assert float(o64.abs().max()) < 1e-6                      # really the tiny regime
    assert rel(out, o64) < 1e-5
    assert rel(zl, z64) < 1e-4
    for (n, _), a, b in zip(conv.named_parameters(), grads, g64):
        assert rel(a, b) < 2e-4, (n, rel(a, b))


def test_backbone_variants_feature_geometry():
    """Swin-T (configs[1]) and Swin-B (configs[3]) backbones: channel counts and the /8, /16, /32
    feature geometry the transformer's level table is built from (odd sizes are padded)."""
    from types import SimpleNamespace

    from ziragroundingdino_amd import backbone as zb

    for name, chans in (("swin_T_224_1k", [192, 384, 768]), ("swin_B_384_22k", [256, 512, 1024])):
        args = SimpleNamespace(backbone=name, hidden_dim=256, pe_temperatureH=20, pe_temperatureW=20,
                               return_interm_indices=[1, 2, 3])
        bb = zb.build_backbone(args).eval()
        assert bb.num_channels == chans
        x = torch.randn(1, 3, 100, 135)
        m = torch.zeros(1, 100, 135, dtype=torch.bool)
        with torch.no_grad():
            feats, poss = bb(utils.NestedTensor(x, m))
        assert [tuple(f.tensors.shape[1:]) for f in feats] == [(chans[0], 13, 17), (chans[1], 7, 9), (chans[2], 4, 5)]
        assert all(p.shape[1] == 256 and p.shape[2:] == f.tensors.shape[2:] for p, f in zip(poss, feats))


@pytest.mark.parametrize("dev", DEVICES)
def test_stacked_criterion_equals_per_set_loop(dev):
    """The criterion's stacked fast path (all 7 prediction sets at once) against its per-set loop,
    which is the reference's structure (two_stage_criterion.py:37-100) and is itself pinned to the
    reference by ``test_two_stage_criterion``: same assignments, same 21 loss values, same
    gradients."""
    from types import SimpleNamespace

    g = torch.Generator().manual_seed(11)
    S, B, Q, C = 7, 2, 40, 32
    logits = (torch.randn(S, B, Q, C, generator=g) * 2).to(dev).requires_grad_(True)
    cxcy = torch.rand(S, B, Q, 2, generator=g) * 0.6 + 0.2
    wh = torch.rand(S, B, Q, 2, generator=g) * 0.3 + 0.05
    boxes = torch.cat([cxcy, wh], -1).to(dev).requires_grad_(True)
    targets = []
    for n in (5, 3):
        c = torch.rand(n, 2, generator=g) * 0.5 + 0.25
        w = torch.rand(n, 2, generator=g) * 0.3 + 0.1
        targets.append({"labels": torch.randint(0, 6, (n,), generator=g).to(dev), "boxes": torch.cat([c, w], -1).to(dev)})
    crit = criterion.build_criterion(SimpleNamespace(aux_loss=True, dec_layers=6, max_text_len=C)).to(dev)
    suffixes = ["_%d" % i for i in range(5)] + ["", "_enc"]
    out = {"pred_logits": logits[5], "pred_boxes": boxes[5],
           "aux_outputs": [{"pred_logits": logits[i], "pred_boxes": boxes[i]} for i in range(5)],
           "enc_outputs": {"pred_logits": logits[6], "pred_boxes": boxes[6]}}
    loop, loop_idx = crit(out, targets, return_indices=True)
    fast, fast_idx = crit(dict(out, stacked=(logits, boxes, suffixes)), targets, return_indices=True)
    assert set(loop) == set(fast) and len(fast) == 21
    # (on the GPU the fast path's assignments come from the device-side solver, the loop's from scipy)
    for (a, b), (c_, d) in zip(loop_idx["indices"] + loop_idx["enc_outputs"][0], fast_idx["indices"] + fast_idx["enc_outputs"][0]):
        assert torch.equal(a.cpu(), c_.cpu()) and torch.equal(b.cpu(), d.cpu())
    for la, fa in zip(loop_idx["aux_outputs"], fast_idx["aux_outputs"]):
        for (a, b), (c_, d) in zip(la, fa):
            assert torch.equal(a.cpu(), c_.cpu()) and torch.equal(b.cpu(), d.cpu())
    if dev == "cuda":
        assert all(i.is_cuda for pair in fast_idx["indices"] for i in pair)
        crit.matcher.check()
    for k in loop:
        close(fast[k], loop[k], 1e-6, k)
    w = crit.weight_dict
    gl = torch.autograd.grad(sum(loop[k] * w[k] for k in loop), [logits, boxes])
    gf = torch.autograd.grad(sum(fast[k] * w[k] for k in fast), [logits, boxes])
    close(gf[0], gl[0], 1e-6, "grad logits")
    close(gf[1], gl[1], 1e-6, "grad boxes")


@pytest.mark.parametrize("dev", DEVICES)
def test_bi_attention_reassociated_equals_reference_order(dev):
    """BiMultiHeadAttention re-brackets its three image-side projections around the (short) text
    side; outputs and input gradients must equal the reference's order of operations
    (``reassociate=False``, itself pinned by the BiAttentionBlock golden vectors) up to fp32
    re-association -- with padding masks on both sides and more than one batch element."""
    torch.manual_seed(5)
    att = transformer.BiMultiHeadAttention(v_dim=64, l_dim=48, embed_dim=128, num_heads=4, dropout=0.0).to(dev)
    for p in att.parameters():
        p.data.normal_(0, 0.2)
    v = torch.randn(2, 700, 64, device=dev, requires_grad=True)
    l = torch.randn(2, 9, 48, device=dev, requires_grad=True)
    mask_v = torch.zeros(2, 700, dtype=torch.bool, device=dev)
    mask_v[1, 600:] = True
    mask_l = torch.zeros(2, 9, dtype=torch.bool, device=dev)
    mask_l[0, 7:] = True
    gv, gl = torch.randn(2, 700, 64, device=dev), torch.randn(2, 9, 48, device=dev)
    res = {}
    for flag in (False, True):
        att.reassociate = flag
        ov, ol = att(v, l, attention_mask_v=mask_v, attention_mask_l=mask_l)
        grads = torch.autograd.grad((ov * gv).sum() + (ol * gl).sum(), [v, l] + list(att.parameters()))
        res[flag] = (ov, ol) + grads
    names = ["out_v", "out_l", "grad_v", "grad_l"] + ["grad " + n for n, _ in att.named_parameters()]
    for n, a, b in zip(names, res[True], res[False]):
        close(a, b, 2e-5, n)


@pytest.mark.parametrize("empty", ["one_image", "all_images"])
def test_stacked_criterion_with_images_without_boxes(empty):
    """Images without ground-truth boxes (common in detection batches): the stacked path must agree
    with the per-set loop there too (class loss only; box losses 0)."""
    from types import SimpleNamespace

    g = torch.Generator().manual_seed(2)
    S, B, Q, C = 7, 2, 30, 16
    logits = torch.randn(S, B, Q, C, generator=g).requires_grad_(True)
    boxes = torch.cat([torch.rand(S, B, Q, 2, generator=g) * 0.6 + 0.2, torch.rand(S, B, Q, 2, generator=g) * 0.3 + 0.05], -1)
    n_per = (0, 4) if empty == "one_image" else (0, 0)
    targets = [{"labels": torch.randint(0, 5, (n,), generator=g),
                "boxes": torch.cat([torch.rand(n, 2, generator=g) * 0.5 + 0.25, torch.rand(n, 2, generator=g) * 0.3 + 0.1], -1)}
               for n in n_per]
    crit = criterion.build_criterion(SimpleNamespace(aux_loss=True, dec_layers=6, max_text_len=C))
    out = {"pred_logits": logits[5], "pred_boxes": boxes[5],
           "aux_outputs": [{"pred_logits": logits[i], "pred_boxes": boxes[i]} for i in range(5)],
           "enc_outputs": {"pred_logits": logits[6], "pred_boxes": boxes[6]}}
    loop = crit(out, targets)
    fast = crit(dict(out, stacked=(logits, boxes, ["_%d" % i for i in range(5)] + ["", "_enc"])), targets)
    assert set(loop) == set(fast)
    for k in loop:
        assert torch.isfinite(fast[k])
        close(fast[k], loop[k], 1e-6, k)


def test_lean_mha_equals_nn_multihead_attention():
    """transformer.lean_mha against nn.MultiheadAttention for the three call patterns of the model:
    self-attention with q = k != v, cross-attention with a key-padding mask, self-attention with a
    per-(batch*head) boolean mask; outputs and input gradients."""
    torch.manual_seed(0)
    mha = torch.nn.MultiheadAttention(64, 4, dropout=0.0).train()
    L, S, B = 11, 7, 3
    x = torch.randn(L, B, 64, requires_grad=True)
    pos = torch.randn(L, B, 64)
    mem = torch.randn(S, B, 64, requires_grad=True)
    kpm = torch.zeros(B, S, dtype=torch.bool)
    kpm[0, 5:] = True
    am = torch.rand(B * 4, L, L) > 0.7
    am[:, torch.arange(L), torch.arange(L)] = False          # never mask a whole row
    qk = x + pos
    cases = [((qk, qk, x), {}), ((x, mem, mem), {"key_padding_mask": kpm}), ((qk, qk, x), {"attn_mask": am})]
    for (q, k, v), kw in cases:
        want = mha(q, k, v, need_weights=False, **kw)[0]
        got = transformer.lean_mha(mha, q, k, v, **kw)
        close(got, want, 2e-6, "output")
        g = torch.randn_like(want)
        wrt = [x] + ([mem] if k is mem else [])
        gw = torch.autograd.grad((want * g).sum(), wrt, retain_graph=True)
        gg = torch.autograd.grad((got * g).sum(), wrt, retain_graph=True)
        for a, b in zip(gg, gw):
            close(a, b, 2e-6, "grad")


def test_bi_attention_block_training_droppath_equals_reference_formula():
    """v + drop_path(gamma_v * delta_v) (reference fuse_modules.py:296-305, timm DropPath) is folded into one
    addcmul with a [B, 1, C] scale: same RNG draws, same values up to one rounding."""
    torch.manual_seed(0)
    blk = transformer.BiAttentionBlock(v_dim=32, l_dim=32, embed_dim=64, num_heads=4, dropout=0.0,
                                       drop_path=0.5, init_values=0.3).train()
    v, l = torch.randn(4, 50, 32), torch.randn(4, 7, 32)
    torch.manual_seed(7)
    got_v, got_l = blk(v, l)
    nv, nl = blk.layer_norm_v(v), blk.layer_norm_l(l)
    dv, dl = blk.attn(nv, nl)
    torch.manual_seed(7)
    want_v = nv + blk.drop_path(blk.gamma_v * dv)
    want_l = nl + blk.drop_path(blk.gamma_l * dl)
    assert torch.allclose(got_v, want_v, atol=1e-6) and torch.allclose(got_l, want_l, atol=1e-6)


def test_small_attention_equals_sdpa():
    """lean_mha's materialised-score path (two batched GEMMs + softmax) against torch SDPA, with an additive
    mask, a broadcast key-padding mask and no mask, values and gradients."""
    g = torch.Generator().manual_seed(2)
    for B, H, L, S, d, kind in [(2, 8, 37, 37, 32, "none"), (2, 4, 11, 11, 64, "full"), (2, 8, 50, 9, 32, "kpm")]:
        q = torch.randn(B, H, L, d, generator=g, requires_grad=True)
        k = torch.randn(B, H, S, d, generator=g, requires_grad=True)
        v = torch.randn(B, H, S, d, generator=g, requires_grad=True)
        mask = None
        if kind == "full":
            mask = torch.zeros(B, H, L, S).masked_fill_(torch.rand(B, H, L, S, generator=g) < 0.3, float("-inf"))
            mask[..., 0] = 0.0                                            # no fully masked row
        elif kind == "kpm":
            mask = torch.zeros(B, 1, 1, S); mask[1, ..., -3:] = float("-inf")
        go = torch.randn(B, H, L, d, generator=g)
        got = transformer._attention_small(q, k, v, mask)
        want = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=mask)
        assert torch.allclose(got, want, atol=2e-6)
        gg = torch.autograd.grad((got * go).sum(), [q, k, v])
        gw = torch.autograd.grad((want * go).sum(), [q, k, v])
        for a, b in zip(gg, gw):
            assert torch.allclose(a, b, atol=5e-6)


def test_sampling_locations_addcdiv_is_bit_identical_on_cpu():
    """One-pass ref + offsets / (W, H) (float divisor) against the reference's expression with its int64
    divisor (ms_deform_attn.py:305-313), values bit for bit and gradients to one rounding."""
    from ziragroundingdino_amd import ms_deform_attn as m
    g = torch.Generator().manual_seed(4)
    ref = torch.rand(2, 60, 4, 2, generator=g)
    off = (torch.randn(2, 60, 8, 4, 4, 2, generator=g) * 3).requires_grad_(True)
    sh = torch.tensor([[100, 167], [50, 84], [25, 42], [13, 21]])
    outs, grads = [], []
    try:
        for fused in (True, False):
            m.FUSED_LOCATIONS = fused
            loc = m.sampling_locations_from_reference_points(ref, off, sh, 4)
            outs.append(loc.detach())
            grads.append(torch.autograd.grad(loc.square().sum(), [off])[0])
    finally:
        m.FUSED_LOCATIONS = True
    assert torch.equal(outs[0], outs[1])
    assert torch.allclose(grads[0], grads[1], rtol=1e-6, atol=1e-8)


# ---------------------------------------------------------------------------------------------
# multilayer-branch ablation (reference groundingdino_dual_zero_rep_multilayer_branch.py:62-226)
def _check_branch_module(mod, g, dev, tol=TOL):
    assert set(mod.state_dict()) == set(g["init_state"])
    for k, v in mod.state_dict().items():           # constructor contract (deterministic parts)
        if "freeze" in k or k == "scaling" or k.startswith("free_") or k == "weight":
            if not ("freeze_self_attn" in k or "freeze_linear1" in k or "freeze_norm" in k):
                close(v, g["init_state"][k], 1e-7, "init " + k)
    mod.load_state_dict(g["state"])
    mod.to(dev).train()
    x = g["x"].clone().requires_grad_(True)
    out, zl = mod(x)
    close(out, g["out"], tol, "out")
    close(zl, g["zl"], tol, "zero-interference loss")
    params = dict(mod.named_parameters())
    grads = torch.autograd.grad((out * g["grad_out"]).sum() + g["zl_weight"] * zl, [x] + list(params.values()))
    close(grads[0], g["grad_x"], tol, "grad x")
    for (k, _), gr in zip(params.items(), grads[1:]):
        close(gr, g["grad_params"][k], tol, "grad " + k)
    mod.eval()
    out_eval, zl_eval = mod(x)
    close(out_eval, g["out_eval"], tol, "eval out")
    assert float(zl_eval) == 0.0
    mod.__rep__()
    for k, v in mod.state_dict().items():
        close(v, g["state_after_rep"][k], tol, "after __rep__ " + k)


@pytest.mark.parametrize("dev", DEVICES)
def test_multilayer_branch_modules(dev):
    from ziragroundingdino_amd import rsb_multilayer as ml

    _check_branch_module(ml.RepZeroLinear(24, 16), to(load("ml_rep_zero_linear"), dev), dev)
    for name in ("1x1", "3x3s2"):
        g = to(load("ml_rep_zero_conv_gn_" + name), dev)
        conv = ml.RepZeroConv2dGN(12, 32, **g["kwargs"])
        assert torch.all(conv.freeze_gn.weight == 1e-8) and torch.all(conv.freeze_gn.bias == 1e-8)
        _check_branch_module(conv, g, dev)
    _check_branch_module(ml.RepZeroTransformerLayer(32, nhead=4, down_dim=48, output_dim=16),
                         to(load("ml_rep_zero_transformer_layer"), dev), dev)


def test_multilayer_branch_model_wiring():
    """Registry name, state-dict names and loss keys of the ablation model (reference :322-323, :384-391, :575-576,
    :665-668): the language branch is ``rep_language_adapter`` / ``loss_language_adapter``, every conv branch carries a
    ``freeze_gn`` that trains (its name contains "adapter") with the 0.2 learning-rate factor ("freeze")."""
    from ziragroundingdino_amd.groundingdino import MODULE_BUILD_FUNCS
    from ziragroundingdino_amd.train import lr_factor
    from test_train_step import build_slice_model, run_slice_step, slice_inputs
    from ziragroundingdino_amd.groundingdino import GroundingDINO

    assert "dualzerorepmultilayerbranchgroundingdino" in MODULE_BUILD_FUNCS
    g = torch.load(os.path.join(GOLDEN, "step_zira_slice.pt"), weights_only=False)

    class Variant(GroundingDINO):
        def __init__(self, *a, **k):
            super().__init__(*a, side_branch="multilayer", **k)

    model = build_slice_model(g, "cpu", Variant)
    names = set(model.state_dict())
    assert "rep_language_adapter.freeze_linear.weight" in names and "rep_linear_adapter.weight" not in names
    assert "input_proj_conv_adapter.3.freeze_gn.weight" in names
    model.before_train()
    trainable = [n for n, p in model.named_parameters() if p.requires_grad]
    assert all("adapter" in n for n in trainable) and "input_proj_conv_adapter.0.freeze_gn.bias" in trainable
    assert lr_factor("input_proj_conv_adapter.0.freeze_gn.bias") == 0.2
    losses = run_slice_step(model, *slice_inputs(g, model, "cpu"))
    assert "loss_language_adapter" in losses and "loss_conv_adapter" in losses and "loss_linear_adapter" not in losses
    sum(losses.values()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for n, p in model.named_parameters() if p.requires_grad)
    model.after_train()
    assert float(model.rep_language_adapter.scaling.detach()) == 1.0
