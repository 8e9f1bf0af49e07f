"""Full-size pin (BASELINE configs[1] hyper-parameters): this package's Transformer + heads, 6 + 6
layers, d = 256, 900 queries, S = 22223, on the GPU through the HIP kernels, against outputs of the
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


def test_full_size_transformer_matches_reference():
    g = torch.load(os.path.join(HERE, "golden", "full_transformer.pt"), weights_only=False)
    tr = attach_heads(transformer.Transformer(**g["kwargs"]), utils.MLP, utils.ContrastiveEmbed)
    assert [n for n, _ in tr.named_parameters()] == g["param_names"]   # state-dict contract at full depth
    fill_by_name_(tr, g["salt"], g["scale"], g["scales"])
    layernorm_weights_plus_one_(tr)
    tr.to("cuda").eval()
    srcs, poss, masks, text, tmask, pid, may, gos = make_inputs()
    dev = lambda x: [t.cuda() for t in x] if isinstance(x, list) else x.cuda()
    srcs = [s.requires_grad_(True) for s in dev(srcs)]
    text = dev(text).requires_grad_(True)
    text_dict = {"encoded_text": text, "text_token_mask": dev(tmask), "position_ids": dev(pid),
                 "text_self_attention_masks": dev(may)}
    hs, refs, hs_enc, ref_enc, init_box, _ = tr(srcs, dev(masks), None, dev(poss), None, None, text_dict)

    # two-stage selection: the same 900 of the 22223 proposals (bit-exact as a set; the fixture's smallest
    # score gap at the cut is 3.5e-3 on a range of 46, adjacent selected ranks can be 1e-5 apart, so the
    # ORDER of near-equal neighbours is allowed to differ and rows are aligned by proposal index below)
    mine, want = tr.last_topk_proposals[0].cpu(), g["topk_proposals"][0]
    assert torch.equal(mine.sort()[0], want.sort()[0])
    pos_of = {int(p): i for i, p in enumerate(mine.tolist())}
    perm = torch.tensor([pos_of[int(p)] for p in want.tolist()], device="cuda")
    assert int((perm != torch.arange(900, device="cuda")).sum()) <= 20   # (a handful of near-tie swaps at most)

    close(text_dict["encoded_text"], g["memory_text"], TOL, "memory_text")
    close(hs[-1][:, perm], g["hs_last"], TOL, "hs[-1]")
    close(hs[0][:, perm][:, ::9], g["hs_first_sample"], TOL, "hs[0] sample")
    close(refs[-1][:, perm], g["reference_last"], TOL, "references[-1]")
    close(hs_enc[:, :, perm][:, :, ::9], g["hs_enc_sample"], TOL, "hs_enc sample")
    close(ref_enc[:, :, perm], g["ref_enc"], TOL, "ref_enc")
    gos = [go.cuda() for go in gos]
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(900, device="cuda")
    total = objective(hs, refs, hs_enc, [go[:, inv] for go in gos])   # grad_out rows follow the query order
    close(total, g["total"], TOL, "objective")
    grads = torch.autograd.grad(total, srcs + [text])
    close(grads[4], g["grad_text"], TOL, "grad text")
    close(torch.stack([x.norm() for x in grads[:4]]), g["grad_src_norms"], TOL, "grad src norms")
    close(grads[3], g["grad_src3"], TOL, "grad srcs[3]")
    close(grads[0][:, ::8, ::10, ::10], g["grad_src0_sample"], TOL, "grad srcs[0] sample")
