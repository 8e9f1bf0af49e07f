"""Full-size pin (BASELINE configs[1] hyper-parameters): this package's Transformer + heads, 6 + 6
layers, d = 256, 900 queries, S = 22223, B = 2 (the benchmarked batch), on the GPU through the HIP kernels, against outputs of the
REFERENCE Transformer run on the CPU of the build container (tests/golden/gen_fullsize_golden.py;
weights rebuilt from parameter names, inputs regenerated from seeds -- the fixture holds compact
outputs only).  north_star: outputs within 1e-3, index selection bit-exact.
"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from gen_fullsize_golden import attach_heads, make_inputs, objective  # noqa: E402
from seeded import fill_by_name_, layernorm_weights_plus_one_  # noqa: E402

from ziragroundingdino_amd import transformer, utils  # noqa: E402

TOL = 1e-3


def close(a, b, tol, what):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    scale = max(1.0, float(b.abs().max()))
    err = float((a - b).abs().max()) / scale
    assert err <= tol, "%s: max err %.3e (scaled by %.3g) > %.1e" % (what, err, scale, tol)


def close_most(a, b, tol, what, frac=0.999, hard=10.0):
    """Elementwise gradients of the full-size model: a sampling location within an ulp of a pixel border falls on
    different sides of `floor` on the two machines, and the piecewise-constant grad_sampling_loc of that ONE sample
    then moves a few elements of a coarse-level gradient by much more than rounding does (measured at B = 2: one
    element of grad srcs[3] at 1.9e-2 of the scale, everything else below 5e-3; the same with round 2's kernels).
    So: `frac` of the elements within `tol`, every element within `hard` x `tol`."""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    scale = max(1.0, float(b.abs().max()))
    err = (a - b).abs() / scale
    ok = float((err <= tol).float().mean())
    assert ok >= frac, "%s: only %.5f of the elements within %.1e" % (what, ok, tol)
    assert float(err.max()) <= hard * tol, "%s: max err %.3e (scaled by %.3g) > %.1e" % (what, float(err.max()), scale, hard * tol)


def count_forward_calls(monkeypatch, fn_class, counts, key):
    """Count the executions of an autograd Function (its ``forward`` static method is looked up on the class at every
    ``apply``)."""
    real = fn_class.forward

    def counted(*a, **k):
        counts[key] = counts.get(key, 0) + 1
        return real(*a, **k)

    monkeypatch.setattr(fn_class, "forward", staticmethod(counted))


@pytest.mark.parametrize("frozen,arith", [(False, "f32"), (True, "f32"), (True, "bf16x3"), (True, "f16x2")],
                         ids=["trainable", "frozen", "frozen_bf16x3", "frozen_f16x2"])
def test_full_size_transformer_matches_reference(monkeypatch, frozen, arith):
    """``frozen``: every parameter with requires_grad = False, as in every ZiRa task (reference
    groundingdino_dual_zero_rep_branch.py:722-745) and in bench.py -- the layers then run as the package's one-node forms
    (decoder_layer.py, encoder_layer.py, the frozen FFN + LayerNorm node, the decoder glue node); with trainable weights those
    nodes decline and the module composition runs.  The fixture's gradients are with respect to the inputs only, so both
    variants are held to the same reference outputs (transformer_for_adapter.py:910-1073, :809-907).
    ``arith`` = "bf16x3": the encoder FFN's four products per layer on the bf16 matrix cores in split-bf16 arithmetic
    (csrc/gemm_bf16x3.hip) -- same bars.  ``arith`` = "f16x2": the encoder FFN as ONE launch per direction on the f16 matrix
    cores (csrc/ffn_f16x2.hip), the other frozen products as under "bf16x3" -- same bars."""
    from ziragroundingdino_amd import decoder_layer, encoder_layer, ffn_f16x2, gemm_bf16x3

    monkeypatch.setattr(transformer.Switches, "gemm_arith", arith)
    real_gemm = gemm_bf16x3.gemm
    split_gemms = []
    monkeypatch.setattr(gemm_bf16x3, "gemm", lambda *a, **k: (split_gemms.append(1), real_gemm(*a, **k))[1])
    real_gemm2, real_gemm3 = gemm_bf16x3.gemm_f16x2, gemm_bf16x3.gemm_f16x2_panel     # (the two-plane f16 forms of the same products, under "f16x2")
    monkeypatch.setattr(gemm_bf16x3, "gemm_f16x2", lambda *a, **k: (split_gemms.append(1), real_gemm2(*a, **k))[1])
    monkeypatch.setattr(gemm_bf16x3, "gemm_f16x2_panel", lambda *a, **k: (split_gemms.append(1), real_gemm3(*a, **k))[1])
    real_ffn, fused_ffns = ffn_f16x2.run, []
    monkeypatch.setattr(ffn_f16x2, "run", lambda *a, **k: (fused_ffns.append(1), real_ffn(*a, **k))[1])

    g = torch.load(os.path.join(HERE, "golden", "full_transformer.pt"), weights_only=False)
    tr = attach_heads(transformer.Transformer(**g["kwargs"]), utils.MLP, utils.ContrastiveEmbed)
    assert [n for n, _ in tr.named_parameters()] == g["param_names"]   # state-dict contract at full depth
    fill_by_name_(tr, g["salt"], g["scale"], g["scales"])
    layernorm_weights_plus_one_(tr)
    tr.to("cuda").eval()
    counts = {}
    for key, cls in (("decoder_layer", decoder_layer._FrozenDecoderLayer), ("decoder_glue", decoder_layer._RefineAndNorm),
                     ("encoder_attention", encoder_layer._FrozenEncoderAttention), ("encoder_ffn", transformer._FrozenFFNNorm)):
        count_forward_calls(monkeypatch, cls, counts, key)
    if frozen:
        for p in tr.parameters():
            p.requires_grad_(False)
    srcs, poss, masks, text, tmask, pid, may, gos = make_inputs()
    dev = lambda x: [t.cuda() for t in x] if isinstance(x, list) else x.cuda()
    srcs = [s.requires_grad_(True) for s in dev(srcs)]
    text = dev(text).requires_grad_(True)
    masks, poss, gos = dev(masks), dev(poss), dev(gos)

    def run():
        text_dict = {"encoded_text": text, "text_token_mask": dev(tmask), "position_ids": dev(pid),
                     "text_self_attention_masks": dev(may)}
        # (frozen: as bench.py's equal-sized images reach the transformer -- the caller knows the masks are all False)
        return tr(srcs, masks, None, poss, None, None, text_dict, no_padding=frozen), text_dict

    # 1. two-stage selection: the same 900 of the 22223 proposals, bit-exact as a set.  The scores of the
    #    selected proposals are 3.5e-3 clear of the 901st (range 46), but neighbours INSIDE the top 900 can be
    #    1e-5 apart, so the order of such near-ties may differ between the CPU's and the GPU's top-k.
    with torch.no_grad():
        run()
    want_all = g["topk_proposals"]
    assert want_all.shape[0] == 2                      # both images of the benchmarked batch
    for b in range(want_all.shape[0]):
        mine, want = tr.last_topk_proposals[b].cpu(), want_all[b]
        assert torch.equal(mine.sort()[0], want.sort()[0])
        moved = (mine != want).nonzero().flatten()
        assert len(moved) <= 40
        srt = g["score_sorted_top1200"][b]
        for i in moved.tolist():   # every displaced entry has a neighbour with a near-equal score
            gap = min(float(srt[i - 1] - srt[i]) if i else 1.0, float(srt[i] - srt[i + 1]))
            assert gap < 1e-4, (b, i, srt[max(i - 2, 0):i + 3])

    # 1b. WITHOUT any help: the package's own top-k order all the way down.  A pair of near-tied proposals that swapped
    #     places wears each other's learnable query embedding (tgt_embed.weight belongs to the POSITION), so those few rows
    #     differ; every other query sees them only through the decoder's self-attention.  Rows whose proposal sits at the
    #     reference's position must match the reference rows (1e-3 of the scale at the median, 2e-2 at most), the
    #     objective stays within 2e-3.
    with torch.no_grad():
        (hs_n, refs_n, hs_enc_n, _, _, _), _ = run()
    same = (tr.last_topk_proposals.cpu() == want_all)                                   # [B, 900]
    scale_h = max(1.0, float(g["hs_last"].abs().max()))
    row_err = ((hs_n[-1].float().cpu() - g["hs_last"]).abs().amax(-1) / scale_h)       # [B, 900]
    assert float(same.float().mean()) >= 1 - 40 / 900
    assert float(row_err[same].median()) <= 1e-3 and float(row_err[same].max()) <= 2e-2, (float(row_err[same].median()), float(row_err[same].max()))
    close(objective(hs_n, refs_n, hs_enc_n, gos), g["total"], 2 * TOL, "objective with the package's own top-k order")

    # 2. everything downstream with the reference's order of those near-ties (a query's initial embedding
    #    belongs to its POSITION, tgt_embed.weight[i], so the pairing position <-> proposal matters)
    real_topk = torch.topk

    def topk_like_reference(x, k, *a, **kw):
        if k == 900 and x.shape[-1] == sum(h * w for h, w in g["shapes"]):
            idx = want_all.to(x.device)
            return torch.gather(x, 1, idx), idx
        return real_topk(x, k, *a, **kw)

    monkeypatch.setattr(torch, "topk", topk_like_reference)
    counts.clear()
    del split_gemms[:]
    del fused_ffns[:]
    (hs, refs, hs_enc, ref_enc, init_box, _), text_dict = run()
    assert torch.equal(tr.last_topk_proposals.cpu(), want_all)
    close(text_dict["encoded_text"], g["memory_text"], TOL, "memory_text")
    close(hs[-1], g["hs_last"], TOL, "hs[-1]")
    close(hs[0][:, ::9], g["hs_first_sample"], TOL, "hs[0] sample")
    close(refs[-1], g["reference_last"], TOL, "references[-1]")
    close(hs_enc[:, :, ::9], g["hs_enc_sample"], TOL, "hs_enc sample")
    close(ref_enc, g["ref_enc"], TOL, "ref_enc")
    total = objective(hs, refs, hs_enc, gos)
    close(total, g["total"], TOL, "objective")
    grads = torch.autograd.grad(total, srcs + [text])
    # Gradients: every forward quantity above is within 1e-3; elementwise gradients of the 12-layer fp32
    # backward (22 223-token reductions folded in different orders on the two machines) are held to 5e-3 of
    # their scale (measured: 2.2e-3 on grad text), their norms to 1e-3.
    GTOL = 5e-3
    close(torch.stack([x.norm() for x in grads[:4]]), g["grad_src_norms"], TOL, "grad src norms")
    close(grads[4].norm(), g["grad_text"].norm(), TOL, "grad text norm")
    close_most(grads[4], g["grad_text"], GTOL, "grad text")
    # (grad srcs[3], 13 x 21 pixels: ONE flipped sample moves ~0.1 % of its elements past 5e-3 -- every variant sits at
    #  0.99903 with the same 1.88e-2 maximum, which is that flip; the split-bf16 arithmetic together with the native text side
    #  flips a second one, 0.99828.  The bar allows a handful of flips; every element stays within 10 x the tolerance.)
    close_most(grads[3], g["grad_src3"], GTOL, "grad srcs[3]", frac=0.999 if arith == "f32" else 0.995)
    close_most(grads[0][:, ::8, ::10, ::10], g["grad_src0_sample"], GTOL, "grad srcs[0] sample")
    # which implementation was pinned by the pass with gradients: six layers each (without gradients the decoder's batched
    # value projections, and with them the one-node layer, stand down: ms_deform_attn.multi_value_projections)
    if frozen:
        assert counts == {"decoder_layer": 6, "decoder_glue": 6, "encoder_attention": 6, "encoder_ffn": 6}, counts
    else:
        assert counts == {}, counts
    # per encoder layer four FFN products and six 256-wide projections of the deformable attention, forward + backward; the six
    # decoder layers' value projections of the memory and their input gradients
    assert len(split_gemms) == {"f32": 0, "bf16x3": 6 * 4 + 6 * 6 + 12, "f16x2": 6 * 6 + 12}[arith]
    assert len(fused_ffns) == (6 + 6 if arith == "f16x2" else 0)   # one launch per encoder layer and direction


def test_swin_b_bf16_training_steps_full_size():
    """BASELINE configs[3]: GroundingDINO-B (Swin-B 384/22k window 12, 1024-channel top level) with the
    GEMMs in bf16 autocast around the fp32 native ops, one 800x1333 image per GPU (bs=8 over DP=8).
    Two trainer steps at full size: every loss finite and fp32, all 25 side-branch tensors receive
    finite non-zero gradients in the flat bucket, and the step moves them."""
    from ziragroundingdino_amd.config import zira_swint_config
    from ziragroundingdino_amd.groundingdino import build_model
    from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch

    torch.manual_seed(0)
    model = build_model(zira_swint_config(device="cuda", backbone="swin_B_384_22k")).to("cuda").train()
    assert list(model.backbone.num_channels) == [256, 512, 1024]
    trainer = ZiraTrainer(model, amp_dtype=torch.bfloat16)
    assert len(trainer.names) == 25
    before = [p.detach().clone() for p in trainer.params]
    data = synthetic_batch(1, 800, 1333, seed=3, device="cuda")
    for it in range(2):
        trainer._check_bucket()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss_dict = model(data)
        for k, v in loss_dict.items():
            assert v.dtype == torch.float32 and torch.isfinite(v), (it, k, v)
        sum(loss_dict.values()).backward()
        for n, p in zip(trainer.names, trainer.params):
            assert torch.isfinite(p.grad).all() and p.grad.abs().max() > 0, (it, n)
        trainer.flat_grad.zero_()
        out = trainer.run_step(data)
        assert set(out) == set(loss_dict)
    moved = [float((p.detach() - b).abs().max()) for p, b in zip(trainer.params, before)]
    assert all(m > 0 for m in moved), dict(zip(trainer.names, moved))


def test_swin_b_bf16_step_tracks_fp32_step():
    """BASELINE configs[3] pinned against fp32: the same GroundingDINO-B weights and minibatch, every source of
    randomness off (dropout, stochastic depth), once in fp32 and once under bf16 autocast.  The summed objective agrees
    to 2e-2, the two zero-interference losses (they sit in front of the top-k query selection) to 1e-3, and the
    gradient the trainer would all-reduce -- the flat side-branch bucket -- has cosine >= 0.99 with its fp32 twin,
    every sizeable tensor of it >= 0.98.  (Single set losses move more: with random-init logits near zero the 900
    selected proposals differ between the two precisions; measured 1e-3 ... 0.5 relative, total 4e-3, cosine 0.9997.)"""
    from torch import nn

    from ziragroundingdino_amd.config import zira_swint_config
    from ziragroundingdino_amd.groundingdino import build_model
    from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch
    from ziragroundingdino_amd.transformer import DropPath

    torch.manual_seed(0)
    model = build_model(zira_swint_config(device="cuda", backbone="swin_B_384_22k")).to("cuda").train()
    model.use_transformer_graph = False
    for m in model.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
        if isinstance(m, DropPath):
            m.drop_prob = 0.0
        if hasattr(m, "_keep_probs"):          # the Swin's pre-drawn stochastic-depth factors
            m._keep_probs.fill_(1.0)
        if hasattr(m, "p_drop"):               # BERT's attention-probability dropout
            m.p_drop = 0.0
    trainer = ZiraTrainer(model, amp_dtype=torch.bfloat16)
    data = synthetic_batch(1, 800, 1333, seed=3, device="cuda")

    def run(bf16):
        trainer.flat_grad.zero_()
        if bf16:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                ld = model(data)
        else:
            ld = model(data)
        sum(ld.values()).backward()
        return {k: float(v) for k, v in ld.items()}, trainer.flat_grad.clone()

    a, ga = run(False)
    a2, ga2 = run(False)
    cos = lambda x, y: float((x * y).sum() / (x.norm() * y.norm() + 1e-30))
    assert cos(ga, ga2) > 0.99999 and all(abs(a[k] - a2[k]) <= 1e-4 * max(1.0, abs(a[k])) for k in a), "fp32 step not repeatable"
    b, gb = run(True)
    assert set(a) == set(b)
    ta, tb = sum(a.values()), sum(b.values())
    assert abs(ta - tb) <= 2e-2 * abs(ta), (ta, tb)
    for k in ("loss_conv_adapter", "loss_linear_adapter"):
        assert abs(a[k] - b[k]) <= 1e-3 * abs(a[k]) + 1e-12, (k, a[k], b[k])
    assert cos(ga, gb) >= 0.99, cos(ga, gb)
    off, total = 0, float(ga.norm())
    for n, p in zip(trainer.names, trainer.params):
        x, y = ga[off:off + p.numel()], gb[off:off + p.numel()]
        off += p.numel()
        if float(x.norm()) > 1e-4 * total:      # (the `scaling` gradients of zero-initialised branches are ~1e-12)
            assert cos(x, y) >= 0.98, (n, cos(x, y))


def test_first_no_grad_forward_of_fresh_frozen_models_selects_the_reference_proposals():
    """Regression (round 5): inside an encoder layer the text enhancer runs on a second stream beside the deformable image
    layer.  Without gradients nothing kept its INPUT alive -- allocated on the main stream, read on the side stream -- and the
    main stream's allocator handed the block to the image layer: 5 of 12 fresh frozen models selected garbage proposals
    (scripts/repro_frozen_nograd.py).  Eight fresh models, first forward each: the reference's 900 proposals as a set."""
    g = torch.load(os.path.join(HERE, "golden", "full_transformer.pt"), weights_only=False)
    srcs, poss, masks, text, tmask, pid, may, _ = make_inputs()
    dev = lambda x: [t.cuda() for t in x] if isinstance(x, list) else x.cuda()
    srcs, poss, masks, text, tmask, pid, may = map(dev, (srcs, poss, masks, text, tmask, pid, may))
    for trial in range(8):
        tr = attach_heads(transformer.Transformer(**g["kwargs"]), utils.MLP, utils.ContrastiveEmbed)
        fill_by_name_(tr, g["salt"], g["scale"], g["scales"])
        layernorm_weights_plus_one_(tr)
        tr.to("cuda").eval()
        for p in tr.parameters():
            p.requires_grad_(False)
        with torch.no_grad():
            tr(srcs, masks, None, poss, None, None, {"encoded_text": text, "text_token_mask": tmask, "position_ids": pid,
                                                     "text_self_attention_masks": may}, no_padding=True)
        for b in range(2):
            assert torch.equal(tr.last_topk_proposals[b].cpu().sort()[0], g["topk_proposals"][b].sort()[0]), (trial, b)
