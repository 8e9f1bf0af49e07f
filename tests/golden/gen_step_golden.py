#!/usr/bin/env python3
"""Golden vectors for the caller harness: two optimisation steps of a
shrunken ZiRa model slice, computed with the REFERENCE's modules on the CPU.

The reference's ``GroundingDINO`` class cannot be constructed offline (it downloads BERT and
needs detectron2), so the slice starts at the tensors its frozen front end would produce --
synthetic Swin feature maps and BERT hidden states -- and re-enacts, with the reference's own
module classes, what ``GroundingDINO.forward`` does from there
(groundingdino_dual_zero_rep_branch.py:459-587): feat_map + RepZeroLinear, input_proj +
RepZeroConv2d + GroupNorm for 4 levels, Transformer, box / class heads, recover_to_cls_logits,
TwoStageCriterion, the 0.1-weighted zero-interference losses; then the driver's step
(train_multidatasets.py:150-200 with test_odinw13_softfreeze/for_train/test_aquarium.py:13-25):
sum of losses, backward, clip_grad_norm_(0.1), AdamW(lr 1e-3, wd 1e-4, x0.2 on "freeze").

Weights are name-seeded (seeded.py) under the parameter names of the full model, nothing big is
stored.      python tests/golden/gen_step_golden.py
"""
import os
import sys
from types import SimpleNamespace

import torch
import torch.nn.functional as F
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
from seeded import fill_by_name_  # noqa: E402

SALT = "zira_step/"
SCALES = {"sampling_offsets.bias": 0.6, "norm": 0.5, "level_embed": 0.3, "input_proj.": 0.08,
          "scaling": 0.3}
CFG = dict(hidden_dim=256, nheads=8, num_queries=16, enc_layers=1, dec_layers=2, dim_feedforward=64,
           num_feature_levels=4, enc_n_points=2, dec_n_points=2, max_text_len=16, bert_hidden=48,
           channels=[12, 20, 28], dropout=0.0)


def build_inputs(g, shapes=((8, 10), (4, 5), (2, 3)), image=(64, 80), pad_from=64):
    """``shapes``: the three backbone levels; ``image``: the padded minibatch size; ``pad_from``: image 1's first padded
    column (None: no padding -- what bench.py's equal-sized images have)."""
    bs = 2
    shapes = [tuple(s) for s in shapes]
    feats = [torch.randn(bs, c, h, w, generator=g) for c, (h, w) in zip(CFG["channels"], shapes)]
    img_mask = torch.zeros(bs, *image, dtype=torch.bool)
    if pad_from is not None:
        img_mask[1, :, pad_from:] = True                         # image 1 is narrower
    masks = [F.interpolate(img_mask[None].float(), size=s).to(torch.bool)[0] for s in shapes]
    poss = [torch.randn(bs, CFG["hidden_dim"], h, w, generator=g) for h, w in shapes]
    h3, w3 = (shapes[-1][0] + 1) // 2, (shapes[-1][1] + 1) // 2   # level 3 = 3x3 s2 conv of level 2
    pos_extra = torch.randn(bs, CFG["hidden_dim"], h3, w3, generator=g)
    ids = torch.tensor([[101, 3000, 1012, 3001, 3002, 1012, 3003, 1012, 102]] * bs)
    hidden = torch.randn(bs, ids.shape[1], CFG["bert_hidden"], generator=g)
    targets = []
    for n in (3, 2):
        c = torch.rand(n, 2, generator=g) * 0.5 + 0.25
        wh = torch.rand(n, 2, generator=g) * 0.3 + 0.1
        targets.append({"labels": torch.randint(0, 3, (n,), generator=g), "boxes": torch.cat([c, wh], -1)})
    return dict(feats=feats, masks=masks, poss=poss, pos_extra=pos_extra, img_mask=img_mask,
                input_ids=ids, bert_hidden=hidden, targets=targets)


class Slice:
    """The reference's modules for the slice, under the parameter names of the full model."""

    def __init__(self, ref, seeded=True):
        Z, T_, U, C = (ref["groundingdino_dual_zero_rep_branch"], ref["transformer_for_adapter"],
                       ref["utils"], ref["criterion"])
        self.ref, self.U = ref, U
        d = CFG["hidden_dim"]
        tr = T_.Transformer(
            d_model=d, nhead=CFG["nheads"], num_queries=CFG["num_queries"], num_encoder_layers=CFG["enc_layers"],
            num_decoder_layers=CFG["dec_layers"], dim_feedforward=CFG["dim_feedforward"], dropout=0.0,
            return_intermediate_dec=True, query_dim=4, num_feature_levels=4, enc_n_points=CFG["enc_n_points"],
            dec_n_points=CFG["dec_n_points"],
            learnable_tgt_init=True, two_stage_type="standard", embed_init_tgt=True, use_text_enhancer=True,
            use_fusion_layer=True, use_text_cross_attention=True, text_dropout=0.0, fusion_dropout=0.0,
            fusion_droppath=0.1, use_adapter=False)
        bbox = U.MLP(d, d, 4, 3)
        cls = U.ContrastiveEmbed(max_text_len=CFG["max_text_len"])
        self.bbox_list = nn.ModuleList([bbox for _ in range(CFG["dec_layers"])])
        tr.decoder.bbox_embed = self.bbox_list
        tr.decoder.class_embed = nn.ModuleList([cls for _ in range(CFG["dec_layers"])])
        tr.enc_out_bbox_embed = U.MLP(d, d, 4, 3)          # two_stage_bbox_embed_share = False
        tr.enc_out_class_embed = cls
        self.tr, self.cls = tr, cls
        self.feat_map = nn.Linear(CFG["bert_hidden"], d)
        self.rep_lin = Z.RepZeroLinear(CFG["bert_hidden"], d)
        chans = CFG["channels"]
        self.input_proj = nn.ModuleList(
            [nn.Sequential(nn.Conv2d(c, d, 1), nn.GroupNorm(32, d)) for c in chans]
            + [nn.Sequential(nn.Conv2d(chans[-1], d, 3, stride=2, padding=1), nn.GroupNorm(32, d))])
        self.adapters = nn.ModuleList([Z.RepZeroConv2d(c, d, kernel_size=1) for c in chans]
                                      + [Z.RepZeroConv2d(chans[-1], d, kernel_size=3, stride=2, padding=1)])
        self.parts = {"transformer.": tr, "feat_map.": self.feat_map, "rep_linear_adapter.": self.rep_lin,
                      "input_proj.": self.input_proj, "input_proj_conv_adapter.": self.adapters}
        if seeded:
            for prefix, m in self.parts.items():
                fill_by_name_(m, SALT, 0.05, SCALES, prefix=prefix)
        for m in self.parts.values():
            m.train()
        for m in (tr, self.feat_map, self.input_proj):        # before_train(): freeze all but "adapter"
            for p in m.parameters():
                p.requires_grad_(False)
        self.crit = C.build_criterion(SimpleNamespace(aux_loss=True, dec_layers=CFG["dec_layers"],
                                                      max_text_len=CFG["max_text_len"]))

    def named_trainable(self):
        return ([("rep_linear_adapter." + n, p) for n, p in self.rep_lin.named_parameters()]
                + [("input_proj_conv_adapter." + n, p) for n, p in self.adapters.named_parameters()])

    def optimizer(self):
        return torch.optim.AdamW(
            [{"params": [p], "lr": 1e-3 * (0.2 if "freeze" in n else 1.0), "weight_decay": 1e-4}
             for n, p in self.named_trainable()], lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-4)

    def state_dict(self):
        return {prefix + n: v for prefix, m in self.parts.items() for n, v in m.state_dict().items()}

    def load_state_dict(self, sd):
        for prefix, m in self.parts.items():
            m.load_state_dict({n[len(prefix):]: v for n, v in sd.items() if n.startswith(prefix)}, strict=True)

    def rep(self):
        for m in list(self.adapters) + [self.rep_lin]:
            m.__rep__()

    def forward(self, inp):
        out, loss_conv, loss_lin = self.outputs(inp)
        crit = self.crit
        loss_dict = crit(out, inp["targets"])
        for k in loss_dict:
            if k in crit.weight_dict:
                loss_dict[k] = loss_dict[k] * crit.weight_dict[k]
        loss_dict["loss_conv_adapter"] = loss_conv * 0.1
        loss_dict["loss_linear_adapter"] = loss_lin * 0.1
        return loss_dict

    def outputs(self, inp):
        """The network part of ``GroundingDINO.forward`` (:459-575): -> (out dict, conv zero loss, linear zero loss)."""
        import groundingdino.util.misc as misc
        U, tr, cls, crit, adapters, input_proj = self.U, self.tr, self.cls, self.crit, self.adapters, self.input_proj
        am, pid, c2t = self.ref["bertwarper"].generate_masks_with_special_tokens_and_transfer_map(
            {"input_ids": inp["input_ids"]}, [101, 102, 1012, 1029], None)
        text_token_mask = torch.ones_like(inp["input_ids"]).bool()
        encoded_text = self.feat_map(inp["bert_hidden"])
        rep_out, loss_lin = self.rep_lin(inp["bert_hidden"])
        encoded_text = rep_out + encoded_text
        text_dict = {"encoded_text": encoded_text, "text_token_mask": text_token_mask,
                     "position_ids": pid, "text_self_attention_masks": am}
        srcs, loss_conv = [], None
        for l, f in enumerate(inp["feats"]):
            a, z = adapters[l](f)
            srcs.append(input_proj[l][1](input_proj[l][0](f) + a))
            loss_conv = z if loss_conv is None else loss_conv + z
        a, z = adapters[3](inp["feats"][-1])
        src3 = input_proj[3][1](input_proj[3][0](inp["feats"][-1]) + a)
        loss_conv = loss_conv + z
        mask3 = F.interpolate(inp["img_mask"][None].float(), size=src3.shape[-2:]).to(torch.bool)[0]
        srcs.append(src3)
        masks = inp["masks"] + [mask3]
        poss = inp["poss"] + [inp["pos_extra"]]
        hs, reference, hs_enc, ref_enc, init_box, _ = tr(srcs, masks, None, poss, None, None, text_dict)
        coords = torch.stack([(bb(h) + misc.inverse_sigmoid(r)).sigmoid()
                              for r, bb, h in zip(reference[:-1], self.bbox_list, hs)])
        classes = torch.stack([U.recover_to_cls_logits(cls(h, text_dict), c2t, for_fill=-100.0) for h in hs])
        out = {"pred_logits": classes[-1], "pred_boxes": coords[-1],
               "aux_outputs": [{"pred_logits": a_, "pred_boxes": b_} for a_, b_ in zip(classes[:-1], coords[:-1])]}
        interm = U.recover_to_cls_logits(tr.enc_out_class_embed(hs_enc[-1], text_dict), c2t, for_fill=-100.0)
        out["enc_outputs"] = {"pred_logits": interm, "pred_boxes": ref_enc[-1]}
        return out, loss_conv, loss_lin


def main():
    torch.set_num_threads(1)
    S = Slice(ref_import.load())
    g = torch.Generator().manual_seed(17)
    inp = build_inputs(g)
    named = S.named_trainable()
    opt = S.optimizer()
    forward = lambda: S.forward(inp)

    steps = []
    for it in range(2):
        loss_dict = forward()
        total = sum(loss_dict.values())
        opt.zero_grad()
        total.backward()
        grads = {n: p.grad.detach().clone() for n, p in named}
        params = [p for _, p in named if p.grad is not None]
        gnorm = torch.nn.utils.clip_grad_norm_(params, max_norm=0.1, norm_type=2)
        opt.step()
        steps.append(dict(loss_dict={k: v.detach().clone() for k, v in loss_dict.items()},
                          total=total.detach().clone(), grad_norm=gnorm.detach().clone(),
                          grads=grads if it == 0 else None,
                          params_after={n: p.detach().clone() for n, p in named} if it == 1 else None))
    # end of task: merge the branches
    S.rep()
    after_rep = {"rep_linear_adapter." + n: v.clone() for n, v in S.rep_lin.state_dict().items()}
    after_rep.update({"input_proj_conv_adapter." + n: v.clone() for n, v in S.adapters.state_dict().items()})
    path = os.path.join(HERE, "step_zira_slice.pt")
    torch.save(dict(cfg=CFG, salt=SALT, scales=SCALES, inputs=inp, steps=steps, after_rep=after_rep,
                    trainable_names=[n for n, _ in named]), path)
    print("step_zira_slice %.1f KiB; losses step0:" % (os.path.getsize(path) / 1024),
          {k: round(float(v), 5) for k, v in steps[0]["loss_dict"].items()})
    print("grad norm", [float(s["grad_norm"]) for s in steps])


if __name__ == "__main__":
    main()
