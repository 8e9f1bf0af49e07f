"""Full-size pin of the transformer (BASELINE configs[1] hyper-parameters) against the REFERENCE.

Runs the reference's `Transformer` (transformer_for_adapter.py:41-415) with the heads its model file
attaches (groundingdino_dual_zero_rep_branch.py:321-361) at the real depth and size -- 6 encoder + 6
decoder layers, d_model 256, FFN 2048, 900 queries, 4 levels of 100x167 / 50x84 / 25x42 / 13x21
(S = 22223), B = 2 (the benchmarked batch), 32 text tokens -- on the CPU of the build container (its pure-PyTorch MSDA path),
forward and backward.  Weights are name-seeded (tests/golden/seeded.py) and inputs are regenerated
from seeds by the test, so the fixture only holds compact outputs: the two-stage top-k indices, the
last decoder layer's hidden states and boxes, a strided sample of the encoder memory, the scalar
objective and gradient norms / samples.  ~2 min of CPU time, ~20 GB of memory.

    python tests/golden/gen_fullsize_golden.py      (needs /root/reference; never runs on the GPU box)
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
from seeded import fill_by_name_, layernorm_weights_plus_one_  # noqa: E402

SALT = "full_transformer/"
# (zero enc_output / enc_output_norm biases: the ~1000 invalid border proposals -- identical all-zero rows,
#  hence one big tie -- then score exactly 0, far below the top 900 of the valid ones)
SCALES = {"sampling_offsets.bias": 0.6, "norm": 0.1, "level_embed": 0.3, "enc_output.bias": 0.0,
          "enc_output_norm.bias": 0.0}
SHAPES = [(100, 167), (50, 84), (25, 42), (13, 21)]
NTOK = 32


def full_args():
    # config/GroundingDINO_SwinT_OGC_rep.py:1-93
    return dict(d_model=256, nhead=8, num_queries=900, num_encoder_layers=6, num_decoder_layers=6,
                dim_feedforward=2048, dropout=0.0, activation="relu", normalize_before=False,
                return_intermediate_dec=True, query_dim=4, num_patterns=0, num_feature_levels=4,
                enc_n_points=4, dec_n_points=4, learnable_tgt_init=True, two_stage_type="standard",
                embed_init_tgt=True, use_text_enhancer=True, use_fusion_layer=True,
                use_text_cross_attention=True, text_dropout=0.0, fusion_dropout=0.0,
                fusion_droppath=0.1, use_adapter=False)


BS = 2   # BASELINE configs[1]: two images per GPU


def make_inputs(d=256, bs=BS):
    """Inputs from fixed seeds (the test calls this too: nothing of it is stored)."""
    g = torch.Generator().manual_seed(77)
    srcs = [torch.randn(bs, d, h, w, generator=g) for h, w in SHAPES]
    poss = [torch.randn(bs, d, h, w, generator=g) for h, w in SHAPES]
    masks = [torch.zeros(bs, h, w, dtype=torch.bool) for h, w in SHAPES]
    text = torch.randn(bs, NTOK, d, generator=g)
    tmask = torch.ones(bs, NTOK, dtype=torch.bool)
    # sub-sentence structure: phrases of 3 tokens + separator, as the mask generator produces
    may = torch.eye(NTOK, dtype=torch.bool).unsqueeze(0).repeat(bs, 1, 1)
    pid = torch.zeros(bs, NTOK, dtype=torch.long)
    for s in range(1, NTOK - 1, 4):
        e = min(s + 4, NTOK - 1)
        may[:, s:e, s:e] = True
        pid[:, s:e] = torch.arange(e - s)
    gos = [torch.randn(bs, 900, d, generator=g) for _ in range(6)]
    return srcs, poss, masks, text, tmask, pid, may, gos


def objective(hs, refs, hs_enc, gos):
    return sum((h * go).sum() for h, go in zip(hs, gos)) + (refs[-1] ** 2).sum() + (hs_enc ** 2).sum() * 0.1


def attach_heads(tr, MLP, ContrastiveEmbed, d=256, layers=6):
    bbox = MLP(d, d, 4, 3)
    cls = ContrastiveEmbed(max_text_len=256)
    tr.decoder.bbox_embed = torch.nn.ModuleList([bbox for _ in range(layers)])
    tr.decoder.class_embed = torch.nn.ModuleList([cls for _ in range(layers)])
    tr.enc_out_bbox_embed = MLP(d, d, 4, 3)
    tr.enc_out_class_embed = cls
    return tr


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    ref = ref_import.load()
    T_, U = ref["transformer_for_adapter"], ref["utils"]
    tr = attach_heads(T_.Transformer(**full_args()), U.MLP, U.ContrastiveEmbed)
    fill_by_name_(tr, SALT, 0.05, SCALES)
    layernorm_weights_plus_one_(tr)
    tr.eval()
    srcs, poss, masks, text, tmask, pid, may, gos = make_inputs()
    srcs = [s.requires_grad_(True) for s in srcs]
    text = text.requires_grad_(True)
    text_dict = {"encoded_text": text, "text_token_mask": tmask, "position_ids": pid,
                 "text_self_attention_masks": may}
    hs, refs, hs_enc, ref_enc, init_box, _ = tr(srcs, masks, None, poss, None, None, dict(text_dict))
    total = objective(hs, refs, hs_enc, gos)
    grads = torch.autograd.grad(total, srcs + [text])
    with torch.no_grad():  # the two-stage selection, recomputed as transformer_for_adapter.py:301-318 does
        src_flat = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
        mask_flat = torch.cat([m.flatten(1) for m in masks], 1)
        pos_flat = torch.cat([p.flatten(2).transpose(1, 2) + tr.level_embed[i].view(1, 1, -1)
                              for i, p in enumerate(poss)], 1)
        sh = torch.tensor(SHAPES)
        lsi = torch.cat((sh.new_zeros((1,)), sh.prod(1).cumsum(0)[:-1]))
        vr = torch.stack([tr.get_valid_ratio(m) for m in masks], 1)
        memory, memory_text, _ = tr.encoder(src_flat, pos=pos_flat, level_start_index=lsi, spatial_shapes=sh,
                                            valid_ratios=vr, key_padding_mask=mask_flat, memory_text=text,
                                            text_attention_mask=~tmask, position_ids=pid,
                                            text_self_attention_masks=may)
        om, _ = U.gen_encoder_output_proposals(memory, mask_flat, sh)
        om_raw = om
        om = tr.enc_output_norm(tr.enc_output(om))
        logits = tr.enc_out_class_embed(om, {"encoded_text": memory_text, "text_token_mask": tmask})
        score = logits.max(-1)[0]
        topk = torch.topk(score, 900, dim=1)[1]
        srt = torch.sort(score, dim=1, descending=True)[0]
    out = dict(
        kwargs=full_args(), salt=SALT, scale=0.05, scales=SCALES, shapes=SHAPES,
        param_names=[n for n, _ in tr.named_parameters()],
        topk_proposals=topk, score_900th_gap=(srt[:, 899] - srt[:, 900]), score_scale=srt[:, 0] - srt[:, -1],
        score_min_gap_top900=(srt[:, :900] - srt[:, 1:901]).min(), score_sorted_top1200=srt[:, :1200].clone(),
        score_of_invalid=score[0][(om_raw.abs().sum(-1) == 0)[0]][:4].clone(), n_invalid=int((om_raw.abs().sum(-1) == 0).sum()),
        memory_sample=memory[:, ::97].clone(), memory_text=memory_text.clone(),
        hs_last=hs[-1].detach().clone(), hs_first_sample=hs[0][:, ::9].detach().clone(),
        reference_last=refs[-1].detach().clone(), hs_enc_sample=hs_enc[:, :, ::9].detach().clone(),
        ref_enc=ref_enc.detach().clone(), total=total.detach(),
        grad_text=grads[4].clone(), grad_src_norms=torch.stack([g.norm() for g in grads[:4]]),
        grad_src3=grads[3].clone(), grad_src0_sample=grads[0][:, ::8, ::10, ::10].clone())
    path = os.path.join(HERE, "full_transformer.pt")
    torch.save(out, path)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1e3), "| total %.6f" % float(total.detach()),
          "| 900th-901st score gap %.3e, smallest gap inside the top 900 %.3e (scale %.3e)"
          % (float(out["score_900th_gap"][0]), float(out["score_min_gap_top900"]), float(out["score_scale"][0])))


if __name__ == "__main__":
    main()
