"""The frozen feed-forward block as one launch per direction on the f16 matrix cores (csrc/ffn_f16x2.hip; reference FFN
transformer_for_adapter.py:877-886 and its autograd under the freeze of groundingdino_dual_zero_rep_branch.py:722-745).
An exact-integer layout check (every fragment permutation shows as a wrong integer), the accuracy gate -- against an fp64
evaluation on the model's own shape the maximum and the rms error must not exceed those of the library's fp32 GEMMs
(what ``F.linear`` runs) --, magnitudes over 24 orders, ragged row counts, the sign bits, determinism and the in-place refresh."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from ziragroundingdino_amd import ffn_f16x2 as ff  # noqa: E402


def _ints(shape, lo, hi, g):
    return torch.randint(lo, hi + 1, shape, device="cuda", generator=g).float()


@pytest.mark.parametrize("M,ws", [(1, True), (100, False), (128, True), (391, True), (391, False), (128 * 300, True)])
def test_exact_on_small_integers(M, ws):
    """Small integers: every product and sum is exact in f16 x f16 -> fp32, so the result must EQUAL the fp32 evaluation --
    with whole 128-row blocks (``ws`` False) and with the last round's row blocks cut into shares of the hidden units whose
    sums meet in the workspace (1, 3 and 4 row blocks: four shares each; 300: 44 row blocks behind a full round, four shares)."""
    g = torch.Generator(device="cuda").manual_seed(1)
    F = 512
    x = _ints((M, 256), -2, 2, g)
    w1 = _ints((F, 256), -1, 1, g) * (torch.rand(F, 256, device="cuda", generator=g) < 0.3)
    w1 = w1 + (torch.arange(F, device="cuda")[:, None] % 3 == 0) * (torch.arange(256, device="cuda")[None, :] % 5 == 0)   # asymmetric
    b1 = _ints((F,), -3, 3, g)
    w2 = _ints((256, F), -1, 1, g) * (torch.rand(256, F, device="cuda", generator=g) < 0.3)
    b2 = _ints((256,), -4, 4, g)
    h = (x @ w1.t() + b1).relu()
    want = h @ w2.t() + b2
    pk = ff.PackedFFN()
    mask = ff.mask_like(x, F)
    got = ff.run(x, pk.get(w1, b1, w2, False), F, False, mask, q_bias=b2, use_workspace=ws)
    assert torch.equal(got, want)
    # backward: gx = aux + ((gy W2) * [h > 0]) W1
    gy = _ints((M, 256), -2, 2, g)
    aux = _ints((M, 256), -5, 5, g)
    want_g = aux + ((gy @ w2) * (h > 0)) @ w1
    got_g = ff.run(gy, pk.get(w1, b1, w2, True), F, True, mask, aux=aux, use_workspace=ws)
    assert torch.equal(got_g, want_g)
    # in place on aux
    acc = aux.clone()
    ff.run(gy, pk.get(w1, b1, w2, True), F, True, mask, aux=acc, out=acc, use_workspace=ws)
    assert torch.equal(acc, want_g)
    if ws:   # the tickets are left zeroed
        n_tail = ((M + 127) // 128) % 256          # (row blocks of the last round on a 256-CU chip)
        tickets = ff.workspace(x.device, torch.cuda.current_stream().cuda_stream, M, F)[:4 * n_tail].view(torch.int32)
        assert int(tickets.abs().sum()) == 0


def _decode_bits(mask, M, F):
    """[M, F] bool from the kernel's private layout: [m][half][step] 16 bits; bit r of (half hf, step c) <-> hidden unit
    32 c + 8 (r / 4) + 4 hf + r % 4."""
    words = mask.view(torch.int16).view(M, 2, F // 32).to(torch.int32) & 0xFFFF
    bits = torch.zeros(M, F, dtype=torch.bool, device=mask.device)
    for hf in range(2):
        for r in range(16):
            bits[:, 8 * (r // 4) + 4 * hf + r % 4::32] = ((words[:, hf, :] >> r) & 1).bool()
    return bits


def _model_like(M, F, seed):
    torch.manual_seed(seed)
    x = torch.randn(M, 256, device="cuda")
    w1 = torch.randn(F, 256, device="cuda") * 0.06
    b1 = torch.randn(F, device="cuda") * 0.1
    w2 = torch.randn(256, F, device="cuda") * 0.03
    b2 = torch.randn(256, device="cuda") * 0.1
    return x, w1, b1, w2, b2


def test_accuracy_gate_against_fp64_beside_the_library_fp32_gemms():
    M, F = 44446, 2048
    x, w1, b1, w2, b2 = _model_like(M, F, 2)
    pk = ff.PackedFFN()
    mask = ff.mask_like(x, F)
    ours = ff.run(x, pk.get(w1, b1, w2, False), F, False, mask, q_bias=b2).double()
    h32 = torch._addmm_activation(b1, x, w1.t())
    lib = torch.addmm(b2, h32, w2.t()).double()
    errs = []
    for lo in range(0, M, 8192):       # fp64 in slices (the [M, F] fp64 activation is 730 MB)
        sl = slice(lo, min(M, lo + 8192))
        ref = torch.addmm(b2.double(), torch.addmm(b1.double(), x[sl].double(), w1.double().t()).relu_(), w2.double().t())
        errs.append(((ours[sl] - ref).abs(), (lib[sl] - ref).abs(), ref.abs().max()))
    scale = float(max(e[2] for e in errs))
    e_ours, e_lib = torch.cat([e[0] for e in errs]), torch.cat([e[1] for e in errs])
    stats = "forward: max %.3e / %.3e, rms %.3e / %.3e of the scale (ours / library)" % (
        float(e_ours.max()) / scale, float(e_lib.max()) / scale, float(e_ours.pow(2).mean().sqrt()) / scale,
        float(e_lib.pow(2).mean().sqrt()) / scale)
    print(stats)
    assert float(e_ours.max()) <= float(e_lib.max()), stats
    assert float(e_ours.pow(2).mean().sqrt()) <= float(e_lib.pow(2).mean().sqrt()), stats
    # backward on the same sign pattern: library = mm, threshold_backward, addmm
    gy = torch.randn(M, 256, device="cuda")
    aux = torch.randn(M, 256, device="cuda")
    ours_g = ff.run(gy, pk.get(w1, b1, w2, True), F, True, mask, aux=aux).double()
    # (on the sign pattern the forward saved: a unit whose pre-activation is within rounding of 0 may differ from fp64's)
    sign = _decode_bits(mask, M, F)
    lib_g = torch.addmm(aux, (gy @ w2) * sign, w1).double()
    eo, el, sc = [], [], 0.0
    for lo in range(0, M, 8192):
        sl = slice(lo, min(M, lo + 8192))
        ref = aux[sl].double() + ((gy[sl].double() @ w2.double()) * sign[sl]) @ w1.double()
        eo.append((ours_g[sl] - ref).abs()); el.append((lib_g[sl] - ref).abs()); sc = max(sc, float(ref.abs().max()))
    eo, el = torch.cat(eo), torch.cat(el)
    stats = "backward: max %.3e / %.3e, rms %.3e / %.3e of the scale (ours / library)" % (
        float(eo.max()) / sc, float(el.max()) / sc, float(eo.pow(2).mean().sqrt()) / sc, float(el.pow(2).mean().sqrt()) / sc)
    print(stats)
    assert float(eo.max()) <= float(el.max()), stats
    assert float(eo.pow(2).mean().sqrt()) <= float(el.pow(2).mean().sqrt()), stats


def test_shares_of_the_last_round_repeat_bit_for_bit():
    """M = 44446: 256 whole row blocks and 92 cut in two; the shares' sums are added in share order by whichever block arrives
    last, so two launches agree bit for bit, and with the whole-block deal to fp32 rounding."""
    M, F = 44446, 2048
    x, w1, b1, w2, b2 = _model_like(M, F, 8)
    pk = ff.PackedFFN()
    mask = ff.mask_like(x, F)
    a = ff.run(x, pk.get(w1, b1, w2, False), F, False, mask, q_bias=b2).clone()
    bits = mask.clone()
    for _ in range(3):
        assert torch.equal(ff.run(x, pk.get(w1, b1, w2, False), F, False, mask, q_bias=b2), a) and torch.equal(mask, bits)
    whole = ff.run(x, pk.get(w1, b1, w2, False), F, False, mask, q_bias=b2, use_workspace=False)
    assert torch.equal(mask, bits)
    assert float((whole - a).abs().max()) <= 2e-6 * float(a.abs().max())
    assert torch.equal(whole[:256 * 128], a[:256 * 128])          # the whole blocks are the same blocks
    gy = torch.randn(M, 256, device="cuda")
    g = ff.run(gy, pk.get(w1, b1, w2, True), F, True, mask, aux=gy).clone()
    for _ in range(3):
        assert torch.equal(ff.run(gy, pk.get(w1, b1, w2, True), F, True, mask, aux=gy), g)


def test_sign_bits_match_the_activation():
    M, F = 300, 512
    x, w1, b1, w2, b2 = _model_like(M, F, 5)
    pk = ff.PackedFFN()
    mask = ff.mask_like(x, F)
    ff.run(x, pk.get(w1, b1, w2, False), F, False, mask, q_bias=b2)
    pre = torch.addmm(b1.double(), x.double(), w1.double().t())
    bits = _decode_bits(mask, M, F)
    sure = pre.abs() > 1e-5
    assert torch.equal(bits[sure], (pre > 0)[sure])
    assert float((~sure).float().mean()) < 1e-3


def test_magnitudes_over_24_orders_and_zero_rows():
    torch.manual_seed(3)
    M, F = 512, 256
    x = torch.randn(M, 256, device="cuda") * torch.logspace(-12, 12, M, device="cuda")[:, None]
    x[7] = 0
    w1, b1 = torch.randn(F, 256, device="cuda"), torch.zeros(F, device="cuda")
    w2 = torch.randn(256, F, device="cuda")
    pk = ff.PackedFFN()
    mask = ff.mask_like(x, F)
    got = ff.run(x, pk.get(w1, b1, w2, False), F, False, mask).double()
    h = (x.double() @ w1.double().t()).relu()
    ref, absref = h @ w2.double().t(), h.abs() @ w2.double().abs().t()
    lib = ((x @ w1.t()).relu() @ w2.t()).double()
    rel = lambda y: float(((y - ref).abs() / absref.clamp_min(1e-300)).max())
    assert rel(got) <= max(rel(lib), 3e-7), (rel(got), rel(lib))
    assert bool((got[7] == 0).all())


def test_rows_past_the_end_are_not_written_and_results_repeat():
    M, F = 200, 256
    x, w1, b1, w2, b2 = _model_like(M, F, 6)
    pk = ff.PackedFFN()
    mask = torch.full((M + 4, F // 32), 0x5A5A5A5A, device="cuda", dtype=torch.int32)
    guard = torch.full((M + 4, 256), 7.0, device="cuda")
    a = ff.run(x, pk.get(w1, b1, w2, False), F, False, mask[:M], q_bias=b2, out=guard[:M]).clone()
    assert bool((guard[M:] == 7.0).all()) and bool((mask[M:] == 0x5A5A5A5A).all())
    b = ff.run(x, pk.get(w1, b1, w2, False), F, False, mask[:M], q_bias=b2)
    assert torch.equal(a, b)


def test_packed_weights_follow_the_parameters_in_place():
    M, F = 256, 256
    x, w1, b1, w2, b2 = _model_like(M, F, 7)
    w1 = torch.nn.Parameter(w1, requires_grad=False)
    pk = ff.PackedFFN()
    p0 = pk.get(w1, b1, w2, False)
    ptr, before = p0.data_ptr(), p0.clone()
    with torch.no_grad():
        w1.mul_(2.0)
    p1 = pk.get(w1, b1, w2, False)
    assert p1.data_ptr() == ptr and not torch.equal(p1, before)
    mask = ff.mask_like(x, F)
    got = ff.run(x, p1, F, False, mask, q_bias=b2)
    want = torch.addmm(b2.double(), torch.addmm(b1.double(), x.double(), w1.double().t()).relu(), w2.double().t())
    assert float((got.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())


def test_argument_errors_are_returned():
    x = torch.randn(8, 256, device="cuda")
    with pytest.raises(RuntimeError):
        ff.pack(torch.randn(100, 256, device="cuda"), torch.zeros(100, device="cuda"), torch.randn(256, 100, device="cuda"), False)
    with pytest.raises(AssertionError):
        ff.run(torch.randn(8, 128, device="cuda"), torch.zeros(16, device="cuda", dtype=torch.uint8), 256, False, ff.mask_like(x, 256))
