"""The decoder layer as one native autograd node (decoder_layer.py: rowgemm prologues / epilogues, attention, MSDA) against the
module composition it replaces (reference transformer_for_adapter.py:910-1073): same output and the same gradients for the
queries, the text memory and the projected image memory, to fp32 re-association."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(Q, B, T, shapes, seed=0, box_refs=True, pad_text=True):
    from ziragroundingdino_amd import transformer
    dev = torch.device("cuda")
    torch.manual_seed(seed)
    layer = transformer.DeformableTransformerDecoderLayer(256, 2048, 0.0, "relu", 4, 8, 4, use_text_cross_attention=True).to(dev).train()
    with torch.no_grad():
        for p in layer.parameters():   # (zero-initialised sampling offsets would hide errors in that path)
            if p.dim() > 1:
                p.normal_(0, 0.05)
            else:
                p.normal_(0, 0.1)
        for n in (layer.norm1, layer.norm2, layer.norm3, layer.catext_norm):
            n.weight.add_(1.0)
    for p in layer.parameters():
        p.requires_grad_(False)
    S = sum(h * w for h, w in shapes)
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    inputs = dict(
        tgt=rnd(Q, B, 256).requires_grad_(True), qpos=rnd(Q, B, 256),
        text=rnd(B, T, 256).requires_grad_(True), value=rnd(B, S, 256).requires_grad_(True),
        memory=rnd(S, B, 256),
        ref=(torch.rand(Q, B, 4, 4 if box_refs else 2, generator=g) * 0.5 + 0.25).to(dev),
        tmask=torch.zeros(B, T, dtype=torch.bool, device=dev),
        shapes=torch.tensor(shapes, device=dev),
        gout=rnd(Q, B, 256),
    )
    if pad_text:
        inputs["tmask"][0, T - 3:] = True
    sh = inputs["shapes"]
    inputs["start"] = torch.cat([sh.new_zeros(1), (sh[:, 0] * sh[:, 1]).cumsum(0)[:-1]])
    return layer, inputs


def _run(layer, x, native):
    layer.native_layer = native
    out = layer(tgt=x["tgt"], tgt_query_pos=x["qpos"], tgt_query_sine_embed=None, tgt_key_padding_mask=None,
                tgt_reference_points=x["ref"], memory_text=x["text"], text_attention_mask=x["tmask"], memory=x["memory"],
                memory_key_padding_mask=None, memory_level_start_index=x["start"], memory_spatial_shapes=x["shapes"],
                memory_pos=None, self_attn_mask=None, cross_attn_mask=None, memory_value=x["value"])[0]
    grads = torch.autograd.grad((out * x["gout"]).sum(), [x["tgt"], x["text"], x["value"]])
    return (out,) + grads


def _rel(a, b):
    a, b = a.detach(), b.detach()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def _bad_row_fraction(a, b, tol):
    """Fraction of rows (last dimension) whose largest error exceeds tol * max|b|."""
    a, b = a.detach(), b.detach()
    bad = (a - b).abs().amax(-1) > tol * float(b.abs().max())
    return float(bad.float().mean())


@pytest.mark.parametrize("Q,B,T,shapes,box_refs", [
    (900, 2, 16, [(25, 34), (13, 17), (7, 9), (4, 5)], True),
    (100, 1, 7, [(12, 10), (6, 5), (3, 3), (2, 2)], True),
    (37, 3, 32, [(16, 16), (8, 8), (4, 4), (2, 2)], False),
])
@pytest.mark.parametrize("frozen_offsets", [False, True])
def test_native_layer_matches_module_composition(Q, B, T, shapes, box_refs, frozen_offsets):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    layer, x = _setup(Q, B, T, shapes, box_refs=box_refs)
    if frozen_offsets:   # offsets = their bias, whatever the query: identical sampling locations on both sides
        with torch.no_grad():
            layer.cross_attn.sampling_offsets.weight.zero_()
            layer.cross_attn.sampling_offsets.bias.uniform_(-3.0, 3.0)
    import ziragroundingdino_amd.decoder_layer as native
    calls = []
    orig = native.decoder_layer_forward
    native.decoder_layer_forward = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        got = _run(layer, x, True)
    finally:
        native.decoder_layer_forward = orig
    assert calls, "the native path did not run"
    ref = _run(layer, x, False)
    for name, a, b in zip(("out", "d tgt", "d text", "d value"), got, ref):
        assert a.shape == b.shape, name
    assert _rel(got[0], ref[0]) < 2e-5
    if frozen_offsets:
        # same sampling locations on both sides: everything agrees to re-association
        for name, a, b in zip(("d tgt", "d text", "d value"), got[1:], ref[1:]):
            assert _rel(a, b) < 2e-4, (name, _rel(a, b))
    else:
        # The two sides form the sampling offsets with GEMMs that add in another order; a location within an ulp of a pixel
        # boundary may then fall into the neighbouring bilinear cell, where the gradient w.r.t. the location is another
        # (the op is not differentiable there): a handful of the 460 k coordinates, i.e. single query rows and the pixels
        # they touch.  Everything else agrees to re-association.
        assert _bad_row_fraction(got[1], ref[1], 2e-4) < 0.005, _bad_row_fraction(got[1], ref[1], 2e-4)
        # (a query row touches up to 512 of the few thousand pixel rows of these small maps)
        assert _bad_row_fraction(got[3], ref[3], 2e-4) < 0.25, _bad_row_fraction(got[3], ref[3], 2e-4)
        assert _rel(got[1], ref[1]) < 0.1 and _rel(got[3], ref[3]) < 0.1
        assert _rel(got[2], ref[2]) < 5e-3, _rel(got[2], ref[2])


def test_native_layer_declines_what_it_does_not_cover():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import ziragroundingdino_amd.decoder_layer as native
    layer, x = _setup(50, 2, 8, [(8, 8), (4, 4), (2, 2), (1, 1)])
    args = (layer, x["tgt"], x["qpos"], x["ref"], x["text"], x["value"], None, None)
    assert native.applies(*args)
    layer.linear1.weight.requires_grad_(True)      # a trainable weight: autograd must see the modules
    assert not native.applies(*args)
    layer.linear1.weight.requires_grad_(False)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert not native.applies(*args)
    assert not native.applies(layer, x["tgt"], x["qpos"], x["ref"], x["text"], x["value"], torch.zeros(50, 50, device="cuda"), None)
    assert not native.applies(layer, x["tgt"], x["qpos"].requires_grad_(True), x["ref"], x["text"], x["value"], None, None)


def test_native_layer_follows_weight_updates_in_place():
    """The transposed weight copies are refreshed in place when a parameter changes (graphs keep reading the same buffers)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    layer, x = _setup(64, 2, 8, [(8, 8), (4, 4), (2, 2), (1, 1)])
    _run(layer, x, True)
    w = layer._native_weights
    ptr = w.sa_in_t.data_ptr()
    with torch.no_grad():
        layer.self_attn.in_proj_weight.mul_(1.5)
        layer.cross_attn.output_proj.weight.add_(0.01)
    layer.refresh_fused_projection()
    assert w.sa_in_t.data_ptr() == ptr
    got = _run(layer, x, True)
    ref = _run(layer, x, False)
    assert _rel(got[0], ref[0]) < 2e-5
    assert _bad_row_fraction(got[1], ref[1], 2e-4) < 0.02 and _rel(got[2], ref[2]) < 5e-3


def _glue_setup(Q=900, B=2, seed=3):
    import types
    from ziragroundingdino_amd.dense import LayerNorm
    from ziragroundingdino_amd.utils import MLP
    dev = torch.device("cuda")
    torch.manual_seed(seed)
    mlp = MLP(256, 256, 4, 3).to(dev)
    norm = LayerNorm(256).to(dev)
    with torch.no_grad():
        norm.weight.normal_(1.0, 0.1)
        norm.bias.normal_(0.0, 0.1)
        mlp.layers[2].weight.normal_(0, 0.05)
    for p in list(mlp.parameters()) + list(norm.parameters()):
        p.requires_grad_(False)
    dec = types.SimpleNamespace(bbox_embed=torch.nn.ModuleList([mlp]), norm=norm)
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = torch.randn(Q, B, 256, generator=g).to(dev).requires_grad_(True)
    ref = torch.rand(Q, B, 4, generator=g).to(dev)
    ref[0, 0] = torch.tensor([0.0, 1.0, 0.0005, 0.9999])     # the clamps of inverse_sigmoid
    g_new = torch.randn(Q, B, 4, generator=g).to(dev)
    g_norm = torch.randn(Q, B, 256, generator=g).to(dev)
    return dec, out, ref, g_new, g_norm


@pytest.mark.parametrize("Q,B", [(900, 2), (37, 3)])
def test_refine_and_norm_matches_the_op_chain(Q, B):
    """Box MLP + inverse_sigmoid + sigmoid and the intermediate LayerNorm as one node (reference
    transformer_for_adapter.py:790-803) against the PyTorch ops, forward and backward, with either output unused."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import ziragroundingdino_amd.decoder_layer as native
    from ziragroundingdino_amd.utils import inverse_sigmoid
    dec, out, ref, g_new, g_norm = _glue_setup(Q, B)
    assert native.refine_applies(dec, 0, out, ref)
    new_ref, normed = native.refine_and_norm(dec, 0, out, ref)
    r_new = (dec.bbox_embed[0](out) + inverse_sigmoid(ref)).sigmoid()
    r_norm = torch.nn.functional.layer_norm(out, (256,), dec.norm.weight, dec.norm.bias, dec.norm.eps)
    assert _rel(new_ref, r_new) < 1e-5 and _rel(normed, r_norm) < 1e-5
    for use_new, use_norm in ((True, True), (True, False), (False, True)):
        outs, refs, gs = [], [], []
        if use_new:
            outs.append(new_ref); refs.append(r_new); gs.append(g_new)
        if use_norm:
            outs.append(normed); refs.append(r_norm); gs.append(g_norm)
        (got,) = torch.autograd.grad(outs, [out], gs, retain_graph=True)
        (want,) = torch.autograd.grad(refs, [out], gs, retain_graph=True)
        assert _rel(got, want) < 2e-5, (use_new, use_norm, _rel(got, want))
    dec.bbox_embed[0].layers[0].weight.requires_grad_(True)
    assert not native.refine_applies(dec, 0, out, ref)


def test_prep_queries_matches_the_op_chain():
    """Boxes per level in both layouts and their sine embedding: bit-identical to the PyTorch ops (reference
    transformer_for_adapter.py:760-770); the position MLP to 1e-5."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import types
    import ziragroundingdino_amd.decoder_layer as native
    from ziragroundingdino_amd.utils import MLP, gen_sineembed_for_position
    dev = torch.device("cuda")
    torch.manual_seed(5)
    head = MLP(512, 256, 256, 2).to(dev)
    for p in head.parameters():
        p.requires_grad_(False)
    dec = types.SimpleNamespace(ref_point_head=head, query_scale=None, query_pos_sine_scale=None)
    for Q, B, L in ((900, 2, 4), (33, 3, 4), (7, 1, 2)):
        ref = torch.rand(Q, B, 4, device=dev)
        ratios = torch.rand(B, L, 2, device=dev) * 0.5 + 0.5
        assert native.prep_applies(dec, ref, ratios)
        ref_in, ref_bf, sine, qpos = native.prep_queries(dec, ref, ratios)
        want_in = ref[:, :, None] * torch.cat([ratios, ratios], -1)[None, :]
        want_sine = gen_sineembed_for_position(want_in[:, :, 0, :])
        assert torch.equal(ref_in, want_in) and torch.equal(ref_bf, want_in.transpose(0, 1).contiguous())
        assert torch.equal(sine, want_sine)
        assert _rel(qpos, head(want_sine)) < 1e-5
    assert not native.prep_applies(dec, ref.requires_grad_(True), ratios)
