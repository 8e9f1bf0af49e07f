#!/usr/bin/env python3
"""Golden vectors for the modules around the MSDA op, generated FROM THE REFERENCE
(its pure-PyTorch CPU path, imported with stubs -- see ref_import.py).  Build container only.

    python tests/golden/gen_modules_golden.py

Writes tests/golden/mod_*.pt (torch.save of plain dicts of tensors / python scalars).
Each file holds the module's state_dict (reference parameter names), seeded inputs, outputs and
-- where the module is differentiable -- gradients w.r.t. inputs and trainable parameters.
"""
import os
import sys
from types import SimpleNamespace

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
from seeded import fill_by_name_  # noqa: E402


def save(name, obj):
    path = os.path.join(HERE, "mod_%s.pt" % name)
    torch.save(obj, path)
    print("%-28s %8.1f KiB" % (name, os.path.getsize(path) / 1024))


def sd(module):
    return {k: v.detach().clone() for k, v in module.state_dict().items()}


def randomize_(module, gen, scale=0.2):
    """Replace the deterministic / zero inits by seeded noise so that every path is exercised."""
    with torch.no_grad():
        for p in module.parameters():
            p.copy_(torch.randn(p.shape, generator=gen) * scale)


def shapes_meta(shapes):
    sh = torch.tensor(shapes, dtype=torch.long)
    start = torch.cat([sh.new_zeros(1), (sh[:, 0] * sh[:, 1]).cumsum(0)[:-1]])
    return sh, start


def gen_msda_module(ref):
    g = torch.Generator().manual_seed(11)
    M = ref["ms_deform_attn"]
    shapes = [(6, 7), (3, 4), (2, 2)]
    sh, start = shapes_meta(shapes)
    S = int((sh[:, 0] * sh[:, 1]).sum())
    for refdim in (2, 4):
        mod = M.MultiScaleDeformableAttention(embed_dim=64, num_heads=4, num_levels=3, num_points=2,
                                              batch_first=True)
        init_sd = sd(mod)  # deterministic init_weights() result for offsets / attention
        randomize_(mod, g, 0.3)
        B, Q = 2, 9
        query = torch.randn(B, Q, 64, generator=g, requires_grad=True)
        value = torch.randn(B, S, 64, generator=g, requires_grad=True)
        qpos = torch.randn(B, Q, 64, generator=g)
        mask = torch.zeros(B, S, dtype=torch.bool)
        mask[1, -3:] = True
        if refdim == 2:
            refp = torch.rand(B, Q, 3, 2, generator=g)
        else:
            refp = torch.cat([torch.rand(B, Q, 3, 2, generator=g), 0.1 + 0.3 * torch.rand(B, Q, 3, 2, generator=g)], -1)
        out = mod(query=query, value=value, query_pos=qpos, key_padding_mask=mask,
                  reference_points=refp, spatial_shapes=sh, level_start_index=start)
        go = torch.randn(out.shape, generator=g)
        params = list(mod.parameters())
        grads = torch.autograd.grad(out, [query, value] + params, go)
        save("msda_module_ref%d" % refdim, dict(
            init_state=init_sd, state=sd(mod), query=query.detach(), value=value.detach(),
            query_pos=qpos, key_padding_mask=mask, reference_points=refp, spatial_shapes=sh,
            level_start_index=start, out=out.detach(), grad_out=go, grad_query=grads[0],
            grad_value=grads[1],
            grad_params={k: gr for (k, _), gr in zip(mod.named_parameters(), grads[2:])}))


def gen_rsb(ref):
    g = torch.Generator().manual_seed(12)
    Z = ref["groundingdino_dual_zero_rep_branch"]
    lin = Z.RepZeroLinear(24, 16)
    init_lin = sd(lin)
    randomize_(lin, g, 0.5)   # large enough that |t| > 1 occurs (both SmoothL1 regimes)
    x = torch.randn(2, 7, 24, generator=g, requires_grad=True) * 3
    lin.train()
    out, zl = lin(x)
    go = torch.randn(out.shape, generator=g)
    total = (out * go).sum() + 0.7 * zl
    params = dict(lin.named_parameters())
    grads = torch.autograd.grad(total, [x] + list(params.values()))
    lin.eval()
    out_eval, zl_eval = lin(x)
    state_before = sd(lin)
    lin.__rep__()
    save("rep_zero_linear", dict(
        init_state=init_lin, state=state_before, x=x.detach(), out=out.detach(), zl=zl.detach(),
        grad_out=go, zl_weight=0.7, grad_x=grads[0],
        grad_params={k: gr for k, gr in zip(params.keys(), grads[1:])},
        out_eval=out_eval.detach(), zl_eval=zl_eval.detach(), state_after_rep=sd(lin)))

    for name, kw in (("1x1", dict(kernel_size=1)), ("3x3s2", dict(kernel_size=3, stride=2, padding=1))):
        conv = Z.RepZeroConv2d(12, 16, **kw)
        init_conv = sd(conv)
        randomize_(conv, g, 0.4)
        x = torch.randn(2, 12, 9, 11, generator=g, requires_grad=True) * 2
        conv.train()
        out, zl = conv(x)
        go = torch.randn(out.shape, generator=g)
        total = (out * go).sum() + 1.3 * zl
        params = dict(conv.named_parameters())
        grads = torch.autograd.grad(total, [x] + list(params.values()))
        conv.eval()
        out_eval, zl_eval = conv(x)
        state_before = sd(conv)
        conv.__rep__()
        save("rep_zero_conv_" + name, dict(
            init_state=init_conv, kwargs=kw, state=state_before, x=x.detach(), out=out.detach(),
            zl=zl.detach(), grad_out=go, zl_weight=1.3, grad_x=grads[0],
            grad_params={k: gr for k, gr in zip(params.keys(), grads[1:])},
            out_eval=out_eval.detach(), zl_eval=zl_eval.detach(), state_after_rep=sd(conv)))


def gen_rsb_multilayer(ref):
    """The multilayer-branch ablation's modules (groundingdino_dual_zero_rep_multilayer_branch.py:62-226)."""
    g = torch.Generator().manual_seed(21)
    Z = ref["groundingdino_dual_zero_rep_multilayer_branch"]

    def record(mod, x, zl_weight, **extra):
        init = sd(mod)
        randomize_(mod, g, 0.4)
        mod.train()
        out, zl = mod(x)
        go = torch.randn(out.shape, generator=g)
        total = (out * go).sum() + zl_weight * zl
        params = dict(mod.named_parameters())
        grads = torch.autograd.grad(total, [x] + list(params.values()))
        mod.eval()
        out_eval, zl_eval = mod(x)
        before = sd(mod)
        mod.__rep__()
        return dict(init_state=init, state=before, x=x.detach(), out=out.detach(), zl=zl.detach(), grad_out=go,
                    zl_weight=zl_weight, grad_x=grads[0], grad_params=dict(zip(params.keys(), grads[1:])),
                    out_eval=out_eval.detach(), zl_eval=zl_eval.detach(), state_after_rep=sd(mod), **extra)

    x = (torch.randn(2, 7, 24, generator=g) * 3).requires_grad_(True)
    save("ml_rep_zero_linear", record(Z.RepZeroLinear(24, 16), x, 0.7))
    for name, kw in (("1x1", dict(kernel_size=1)), ("3x3s2", dict(kernel_size=3, stride=2, padding=1))):
        x = (torch.randn(2, 12, 9, 11, generator=g) * 2).requires_grad_(True)
        save("ml_rep_zero_conv_gn_" + name, record(Z.RepZeroConv2dGN(12, 32, **kw), x, 1.3, kwargs=kw))
    x = torch.randn(9, 2, 32, generator=g).requires_grad_(True)
    save("ml_rep_zero_transformer_layer", record(Z.RepZeroTransformerLayer(32, nhead=4, down_dim=48, output_dim=16), x, 0.9))


def fake_tokens(gen, bs, caps):
    """input_ids in BERT style: [CLS]=101 words... '.'=1012 ... [SEP]=102, right-padded with 0."""
    rows = []
    for words_per_cat in caps:
        ids = [101]
        for n in words_per_cat:
            ids += torch.randint(2000, 20000, (n,), generator=gen).tolist() + [1012]
        ids.append(102)
        rows.append(ids)
    T = max(len(r) for r in rows)
    input_ids = torch.zeros(bs, T, dtype=torch.long)
    attn = torch.zeros(bs, T, dtype=torch.long)
    for i, r in enumerate(rows):
        input_ids[i, :len(r)] = torch.tensor(r)
        attn[i, :len(r)] = 1
    return input_ids, attn


SPECIAL = [101, 102, 1012, 1029]


def gen_text_and_logits(ref):
    g = torch.Generator().manual_seed(13)
    B = ref["bertwarper"]
    U = ref["utils"]
    # In the reference every image of a batch carries the same caption (all categories of the
    # dataset), so [SEP] always sits in the last column; unequal lengths are only exercised on
    # the mask generator itself (a mid-row [SEP] yields an EMPTY category mask, on which the
    # reference's recover_to_cls_logits raises).
    input_ids, attn = fake_tokens(g, 2, [[1, 2, 1, 3], [2, 1, 3, 1]])
    tok = {"input_ids": input_ids, "attention_mask": attn}
    am, pid, c2t = B.generate_masks_with_special_tokens_and_transfer_map(tok, SPECIAL, None)
    ids3, _ = fake_tokens(g, 2, [[1, 2, 1, 3], [2, 1, 1]])
    am3, pid3, c2t3 = B.generate_masks_with_special_tokens_and_transfer_map({"input_ids": ids3}, SPECIAL, None)
    # also a '?' separated caption and a single-category one
    ids2 = torch.tensor([[101, 5000, 5001, 1029, 5002, 1012, 102], [101, 7000, 1012, 102, 0, 0, 0]])
    am2, pid2, c2t2 = B.generate_masks_with_special_tokens_and_transfer_map({"input_ids": ids2}, SPECIAL, None)
    save("text_masks", dict(input_ids=input_ids, attention_mask=am, position_ids=pid, cate_to_token=c2t,
                            special=SPECIAL, input_ids2=ids2, attention_mask2=am2, position_ids2=pid2,
                            cate_to_token2=c2t2, input_ids3=ids3, attention_mask3=am3,
                            position_ids3=pid3, cate_to_token3=c2t3))

    T = input_ids.shape[1]
    hs = torch.randn(2, 15, 32, generator=g, requires_grad=True)
    text = torch.randn(2, T, 32, generator=g, requires_grad=True)
    text_dict = {"encoded_text": text, "text_token_mask": attn.bool()}
    ce = U.ContrastiveEmbed(max_text_len=24)
    logits = ce(hs, text_dict)
    cls = U.recover_to_cls_logits(logits, c2t, for_fill=-100.0)
    go = torch.randn(cls.shape, generator=g)
    ghs, gtext = torch.autograd.grad(cls, [hs, text], go)
    save("contrastive_logits", dict(hs=hs.detach(), text=text.detach(), text_token_mask=attn.bool(),
                                    cate_to_token=c2t, max_text_len=24, token_logits=logits.detach(),
                                    cls_logits=cls.detach(), grad_out=go, grad_hs=ghs, grad_text=gtext))

    pos4 = torch.rand(5, 2, 4, generator=g)
    pos1 = torch.randint(0, 9, (2, 6, 1), generator=g).float()
    mem = torch.randn(2, 6 * 7 + 3 * 4, 8, generator=g)
    pad = torch.zeros(2, 6 * 7 + 3 * 4, dtype=torch.bool)
    pad[1].view(-1)[6 * 7 - 7:6 * 7] = True          # last row of level 0 padded for image 1
    pad[1, -4:] = True                                # last row of level 1
    om, op = U.gen_encoder_output_proposals(mem, pad, torch.tensor([[6, 7], [3, 4]]))
    import groundingdino.util.misc as misc
    x = torch.tensor([-0.5, 0.0, 1e-4, 0.3, 0.999, 1.0, 1.7])
    save("utils_misc", dict(pos4=pos4, sine4=U.gen_sineembed_for_position(pos4),
                            sine2=U.gen_sineembed_for_position(pos4[..., :2]),
                            pos1=pos1, sine1=U.get_sine_pos_embed(pos1, num_pos_feats=16, exchange_xy=False),
                            memory=mem, padding_mask=pad, shapes=[(6, 7), (3, 4)], out_memory=om,
                            out_proposals=op, inv_sig_in=x, inv_sig_out=misc.inverse_sigmoid(x)))


def gen_criterion(ref):
    g = torch.Generator().manual_seed(14)
    C = ref["criterion"]
    args = SimpleNamespace(aux_loss=True, dec_layers=3, max_text_len=16)
    crit = C.build_criterion(args)
    bs, nq, K = 2, 30, 16
    n_cat = [5, 3]

    def fake_logits():
        lg = torch.randn(bs, nq, K, generator=g) * 2
        for b in range(bs):
            lg[b, :, n_cat[b]:] = -100.0        # recover_to_cls_logits fill
        return lg.requires_grad_(True)

    def fake_boxes():
        c = torch.rand(bs, nq, 2, generator=g) * 0.6 + 0.2
        wh = torch.rand(bs, nq, 2, generator=g) * 0.3 + 0.05
        return torch.cat([c, wh], -1).requires_grad_(True)

    out = {"pred_logits": fake_logits(), "pred_boxes": fake_boxes(),
           "aux_outputs": [{"pred_logits": fake_logits(), "pred_boxes": fake_boxes()} for _ in range(2)],
           "enc_outputs": {"pred_logits": fake_logits(), "pred_boxes": fake_boxes()}}
    targets = []
    for b, n in enumerate([4, 6]):
        c = torch.rand(n, 2, generator=g) * 0.6 + 0.2
        wh = torch.rand(n, 2, generator=g) * 0.3 + 0.05
        targets.append({"labels": torch.randint(0, n_cat[b], (n,), generator=g), "boxes": torch.cat([c, wh], -1)})
    losses, idx = crit(out, targets, return_indices=True)
    total = sum(losses[k] * crit.weight_dict[k] for k in losses)
    leaves = [out["pred_logits"], out["pred_boxes"]]
    grads = torch.autograd.grad(total, leaves)
    cost = crit.matcher
    strip = lambda d: {k: v.detach() for k, v in d.items()}
    save("criterion", dict(
        args=vars(args), outputs=dict(pred_logits=out["pred_logits"].detach(), pred_boxes=out["pred_boxes"].detach(),
                                      aux_outputs=[strip(a) for a in out["aux_outputs"]],
                                      enc_outputs=strip(out["enc_outputs"])),
        targets=targets, losses={k: v.detach() for k, v in losses.items()}, weight_dict=dict(crit.weight_dict),
        indices=idx, total=total.detach(), grad_pred_logits=grads[0], grad_pred_boxes=grads[1]))


def gen_encoder_neighbours(ref):
    g = torch.Generator().manual_seed(15)
    F_ = ref["fuse_modules"]
    V = ref["transformer_vanilla"]
    blk = F_.BiAttentionBlock(v_dim=32, l_dim=32, embed_dim=64, num_heads=4, dropout=0.0, drop_path=0.1)
    randomize_(blk, g, 0.3)
    blk.eval()   # the DropPath stub is identity anyway
    v = torch.randn(2, 20, 32, generator=g, requires_grad=True)
    l = torch.randn(2, 7, 32, generator=g, requires_grad=True)
    mv = torch.zeros(2, 20, dtype=torch.bool); mv[1, -4:] = True
    ml = torch.zeros(2, 7, dtype=torch.bool); ml[0, -2:] = True
    ov, ol = blk(v, l, attention_mask_v=mv, attention_mask_l=ml)
    gov, gol = torch.randn(ov.shape, generator=g), torch.randn(ol.shape, generator=g)
    gv, gl = torch.autograd.grad((ov * gov).sum() + (ol * gol).sum(), [v, l])
    save("bi_attention_block", dict(state=sd(blk), v=v.detach(), l=l.detach(), mask_v=mv, mask_l=ml,
                                    out_v=ov.detach(), out_l=ol.detach(), grad_out_v=gov, grad_out_l=gol,
                                    grad_v=gv, grad_l=gl))

    lay = V.TransformerEncoderLayer(d_model=32, nhead=4, dim_feedforward=48, dropout=0.0)
    randomize_(lay, g, 0.3)
    lay.eval()
    T = 7
    src = torch.randn(T, 2, 32, generator=g, requires_grad=True)
    pos = torch.randn(T, 2, 32, generator=g)
    # block-diagonal "may attend" masks, different per image (exposes the head/batch tiling quirk)
    may = torch.eye(T, dtype=torch.bool).unsqueeze(0).repeat(2, 1, 1)
    may[0, 1:4, 1:4] = True
    may[1, 1:3, 1:3] = True
    may[1, 3:6, 3:6] = True
    out = lay(src, src_mask=~may, src_key_padding_mask=None, pos=pos)
    go = torch.randn(out.shape, generator=g)
    gs, = torch.autograd.grad(out, [src], go)
    save("text_enhancer_layer", dict(state=sd(lay), src=src.detach(), pos=pos, may_attend=may,
                                     out=out.detach(), grad_out=go, grad_src=gs))


def tiny_transformer_args():
    # d_model must be 256: the encoder hard-codes 256 sine features for the text positions
    # (transformer_for_adapter.py:548-557)
    return dict(d_model=256, nhead=8, num_queries=12, num_encoder_layers=2, num_decoder_layers=2,
                dim_feedforward=64, dropout=0.0, activation="relu", normalize_before=False,
                return_intermediate_dec=True, query_dim=4, num_patterns=0, num_feature_levels=3,
                enc_n_points=2, dec_n_points=2, learnable_tgt_init=True, two_stage_type="standard",
                embed_init_tgt=True, use_text_enhancer=True, use_fusion_layer=True,
                use_text_cross_attention=True, text_dropout=0.0, fusion_dropout=0.0,
                fusion_droppath=0.1, use_adapter=False)


TRANSFORMER_SALT = "tiny_transformer/"
TRANSFORMER_SCALES = {"sampling_offsets.bias": 0.6, "norm": 0.5, "level_embed": 0.3}


def gen_transformer(ref):
    g = torch.Generator().manual_seed(16)
    T_ = ref["transformer_for_adapter"]
    U = ref["utils"]
    kw = tiny_transformer_args()
    d = kw["d_model"]
    tr = T_.Transformer(**kw)
    # heads the model file attaches (groundingdino_dual_zero_rep_branch.py:321-361)
    bbox = U.MLP(d, d, 4, 3)
    cls = U.ContrastiveEmbed(max_text_len=16)
    tr.decoder.bbox_embed = torch.nn.ModuleList([bbox for _ in range(2)])
    tr.decoder.class_embed = torch.nn.ModuleList([cls for _ in range(2)])
    tr.enc_out_bbox_embed = U.MLP(d, d, 4, 3)
    tr.enc_out_class_embed = cls
    fill_by_name_(tr, TRANSFORMER_SALT, 0.05, TRANSFORMER_SCALES)   # weights are NOT stored
    tr.eval()  # dropout 0 and the DropPath stub is identity

    shapes = [(6, 8), (3, 4), (2, 2)]
    bs = 2
    srcs = [torch.randn(bs, d, h, w, generator=g, requires_grad=True) for h, w in shapes]
    poss = [torch.randn(bs, d, h, w, generator=g) for h, w in shapes]
    masks = []
    for h, w in shapes:
        m = torch.zeros(bs, h, w, dtype=torch.bool)
        m[1, :, -(w // 4 or 1):] = True     # image 1 is narrower: right columns are padding
        masks.append(m)
    ntok = 6
    may = torch.eye(ntok, dtype=torch.bool).unsqueeze(0).repeat(bs, 1, 1)
    may[:, 1:3, 1:3] = True
    may[:, 3:5, 3:5] = True
    text = torch.randn(bs, ntok, d, generator=g, requires_grad=True)
    tmask = torch.ones(bs, ntok, dtype=torch.bool)
    tmask[1, -1] = False
    pid = torch.tensor([[0, 0, 1, 0, 1, 0]] * bs)
    text_dict = {"encoded_text": text, "text_token_mask": tmask, "position_ids": pid,
                 "text_self_attention_masks": may}
    hs, refs, hs_enc, ref_enc, init_box, aloss = tr(srcs, masks, None, poss, None, None, dict(text_dict))
    gos = [torch.randn(h.shape, generator=g) for h in hs]
    total = sum((h * go).sum() for h, go in zip(hs, gos)) + (refs[-1] ** 2).sum() + (hs_enc ** 2).sum() * 0.1
    grads = torch.autograd.grad(total, srcs + [text])
    # two-stage top-k indices, recomputed exactly as transformer_for_adapter.py:301-318 does
    with torch.no_grad():
        src_flat = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
        mask_flat = torch.cat([m.flatten(1) for m in masks], 1)
        pos_flat = torch.cat([p.flatten(2).transpose(1, 2) + tr.level_embed[i].view(1, 1, -1)
                              for i, p in enumerate(poss)], 1)
        sh = torch.tensor(shapes)
        lsi = torch.cat((sh.new_zeros((1,)), sh.prod(1).cumsum(0)[:-1]))
        vr = torch.stack([tr.get_valid_ratio(m) for m in masks], 1)
        memory, memory_text, _ = tr.encoder(src_flat, pos=pos_flat, level_start_index=lsi, spatial_shapes=sh,
                                            valid_ratios=vr, key_padding_mask=mask_flat, memory_text=text,
                                            text_attention_mask=~tmask, position_ids=pid,
                                            text_self_attention_masks=may)
        om, op = U.gen_encoder_output_proposals(memory, mask_flat, sh)
        om = tr.enc_output_norm(tr.enc_output(om))
        logits = tr.enc_out_class_embed(om, {"encoded_text": memory_text, "text_token_mask": tmask})
        topk = torch.topk(logits.max(-1)[0], kw["num_queries"], dim=1)[1]
    save("tiny_transformer", dict(
        kwargs=kw, salt=TRANSFORMER_SALT, scale=0.05, scales=TRANSFORMER_SCALES,
        param_names=[n for n, _ in tr.named_parameters()], shapes=shapes,
        srcs=[s.detach() for s in srcs], poss=poss, masks=masks,
        text=text.detach(), text_token_mask=tmask, position_ids=pid, text_self_attention_masks=may,
        memory=memory, memory_text=memory_text, topk_proposals=topk,
        hs=[h.detach() for h in hs], references=[r.detach() for r in refs], hs_enc=hs_enc.detach(),
        ref_enc=ref_enc.detach(), init_box_proposal=init_box.detach(), grad_hs=gos,
        grad_srcs=list(grads[:3]), grad_text=grads[3], total=total.detach()))


def main():
    torch.set_num_threads(1)
    ref = ref_import.load()
    gen_msda_module(ref)
    gen_rsb(ref)
    gen_rsb_multilayer(ref)
    gen_text_and_logits(ref)
    gen_criterion(ref)
    gen_encoder_neighbours(ref)
    gen_transformer(ref)


if __name__ == "__main__":
    main()
