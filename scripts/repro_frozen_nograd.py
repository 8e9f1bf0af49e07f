"""Dev: hunt for the intermittent garbage in the FIRST no-grad forward of the frozen full-size transformer
(tests/test_fullsize_gpu.py [frozen]).  Fresh model per trial (the failure was seen on first use), N trials per switch setting."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from gen_fullsize_golden import attach_heads, make_inputs
from seeded import fill_by_name_, layernorm_weights_plus_one_
from ziragroundingdino_amd import transformer, utils

g = torch.load(os.path.join(ROOT, "tests", "golden", "full_transformer.pt"), weights_only=False)
srcs, poss, masks, text, tmask, pid, may, gos = make_inputs()
dev = lambda x: [t.cuda() for t in x] if isinstance(x, list) else x.cuda()
srcs, poss, masks, text, tmask, pid, may = map(dev, (srcs, poss, masks, text, tmask, pid, may))
want = g["topk_proposals"]


def trial(frozen, prerun_backward):
    tr = attach_heads(transformer.Transformer(**g["kwargs"]), utils.MLP, utils.ContrastiveEmbed)
    fill_by_name_(tr, g["salt"], g["scale"], g["scales"])
    layernorm_weights_plus_one_(tr)
    tr.to("cuda").eval()
    if frozen:
        for p in tr.parameters():
            p.requires_grad_(False)
    td = lambda: {"encoded_text": text, "text_token_mask": tmask, "position_ids": pid, "text_self_attention_masks": may}
    with torch.no_grad():
        hs, refs, hs_enc, ref_enc, init_box, _ = tr(srcs, masks, None, poss, None, None, td(), no_padding=frozen)
    ok = all(torch.equal(tr.last_topk_proposals[b].cpu().sort()[0], want[b].sort()[0]) for b in range(2))
    fin = bool(torch.isfinite(hs_enc).all())
    return ok, fin


for name, setup in (("default", lambda: None),
                    ("no text overlap", lambda: setattr(transformer.TransformerEncoder, "overlap_text_layer", False)),
                    ("no composed text side", lambda: (setattr(transformer.TransformerEncoder, "overlap_text_layer", True),
                                                       setattr(transformer.BiMultiHeadAttention, "compose_text_side", False)))):
    setup()
    bad = 0
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    for i in range(n):
        ok, fin = trial(True, False)
        bad += 0 if ok else 1
        if not ok:
            print("  trial %d: selection wrong, hs_enc finite=%s" % (i, fin), flush=True)
    print("%-24s %d / %d trials wrong" % (name, bad, n), flush=True)
