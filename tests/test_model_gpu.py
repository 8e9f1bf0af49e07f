"""Whole-model checks on the GPU: the ZiRa GroundingDINO model (small Swin / BERT so that it runs in
seconds) trains through the HIP kernels, the hipGraph replay of the frozen front end returns what
the eager front end returns, eval mode produces detections, and only the side branches move."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

from ziragroundingdino_amd import backbone as zb  # noqa: E402
from ziragroundingdino_amd import bert as zbert  # noqa: E402
from ziragroundingdino_amd.config import zira_swint_config  # noqa: E402
from ziragroundingdino_amd.criterion import build_criterion  # noqa: E402
from ziragroundingdino_amd.groundingdino import GroundingDINO  # noqa: E402
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch  # noqa: E402
from ziragroundingdino_amd.transformer import build_transformer  # noqa: E402


def small_model(dev="cuda", drop_path_rate=0.0):
    torch.manual_seed(0)
    args = zira_swint_config(num_queries=50, enc_layers=2, dec_layers=2, dim_feedforward=128,
                             fusion_droppath=0.0, max_text_len=32)
    swin = zb.SwinTransformer(embed_dim=24, depths=(1, 1, 2, 1), num_heads=(1, 2, 4, 8), window_size=7,
                              drop_path_rate=drop_path_rate, out_indices=(1, 2, 3))
    bb = zb.Joiner(swin, zb.PositionEmbeddingSineHW(128, 20, 20, normalize=True))
    bb.num_channels = swin.num_features[1:]
    # (no dropout in the text encoder: several tests compare two forward passes of a model in training mode)
    tiny_bert = zbert.BertModel(zbert.BertConfig(hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                                                 intermediate_size=128, hidden_dropout_prob=0.0,
                                                 attention_probs_dropout_prob=0.0))
    model = GroundingDINO(bb, build_transformer(args), num_queries=50, aux_loss=True, iter_update=True,
                          query_dim=4, num_feature_levels=4, nheads=8, two_stage_type="standard",
                          two_stage_bbox_embed_share=False, two_stage_class_embed_share=False,
                          max_text_len=32, criterion=build_criterion(args), freeze_all=True, use_cet=True,
                          use_project_adapter=True, device=dev, bert=tiny_bert,
                          select_box_nums_for_evaluation=20)
    return model.to(dev)


def test_training_steps_move_only_side_branches_and_reduce_loss():
    model = small_model().train()
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    trainer = ZiraTrainer(model)
    assert all("adapter" in n for n in trainer.names) and len(trainer.names) == 25
    data = synthetic_batch(2, 224, 320, n_categories=4, boxes_per_image=3, device="cuda")
    losses = []
    for _ in range(6):
        out = trainer.run_step(data)
        losses.append(float(sum(out.values())))
        assert all(torch.isfinite(v) for v in out.values())
    assert {"loss_class", "loss_bbox", "loss_giou", "loss_class_0", "loss_class_enc",
            "loss_conv_adapter", "loss_linear_adapter"} <= set(out)
    assert losses[-1] < losses[0]
    moved = [n for n, p in model.named_parameters() if not torch.equal(p.detach(), before[n])]
    assert moved and all("adapter" in n for n in moved)


def test_frontend_graph_replay_matches_eager():
    model = small_model().train()
    model.before_train()
    data = synthetic_batch(2, 224, 320, n_categories=4, boxes_per_image=3, device="cuda")
    from ziragroundingdino_amd.utils import nested_tensor_from_tensor_list

    samples = nested_tensor_from_tensor_list(model.preprocess_image(data))
    model.use_frontend_graphs = False
    f_eager, p_eager = model.run_backbone(samples)
    f_eager = [f.tensors.clone() for f in f_eager]
    model.use_frontend_graphs = True
    for _ in range(3):                       # capture, then two replays
        f_graph, p_graph = model.run_backbone(samples)
        for a, b in zip(f_eager, f_graph):
            torch.testing.assert_close(b.tensors, a, rtol=1e-4, atol=1e-4)
        for a, b in zip(p_eager, p_graph):
            torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-5)
    loss_g = model(data)
    model.use_frontend_graphs = False
    loss_e = model(data)
    for k in loss_e:
        torch.testing.assert_close(loss_g[k], loss_e[k], rtol=2e-4, atol=2e-4)


def test_frontend_graph_follows_mode_and_outputs_survive_replays():
    """The captured front end must not leak across modes: stochastic depth (drop_path > 0) only runs
    in training, so a graph captured while training is not the one replayed after .eval(), and eval
    results are deterministic and equal to the eager front end.  The features handed out are copies:
    a later replay (gradient accumulation, an eval pass before backward) leaves them intact."""
    from ziragroundingdino_amd.utils import nested_tensor_from_tensor_list

    model = small_model(drop_path_rate=0.3).train()
    model.before_train()
    data = synthetic_batch(2, 224, 320, n_categories=4, boxes_per_image=3, device="cuda")
    samples = nested_tensor_from_tensor_list(model.preprocess_image(data))
    model.use_frontend_graphs = True
    f_train, _ = model.run_backbone(samples)            # capture in training mode (drop path active)
    f_train = [f.tensors for f in f_train]
    kept = [t.clone() for t in f_train]
    f_train2, _ = model.run_backbone(samples)           # another replay: different drop-path draws
    assert any(not torch.equal(a.tensors, b) for a, b in zip(f_train2, kept))
    for a, b in zip(f_train, kept):                     # ... and the earlier outputs were not overwritten
        assert torch.equal(a, b)
    model.eval()
    g1, _ = model.run_backbone(samples)
    g2, _ = model.run_backbone(samples)
    model.use_frontend_graphs = False
    e, _ = model.run_backbone(samples)
    for a, b, c in zip(g1, g2, e):
        assert torch.equal(a.tensors, b.tensors)        # eval is deterministic under replay
        torch.testing.assert_close(a.tensors, c.tensors, rtol=1e-4, atol=1e-4)
    # autocast state is part of the key: an fp32 capture is not replayed under bf16 autocast
    model.use_frontend_graphs = True
    with torch.autocast("cuda", dtype=torch.bfloat16):
        h, _ = model.run_backbone(samples)
    model.use_frontend_graphs = False
    with torch.autocast("cuda", dtype=torch.bfloat16):
        he, _ = model.run_backbone(samples)
    for a, b in zip(h, he):
        assert a.tensors.dtype == b.tensors.dtype
        torch.testing.assert_close(a.tensors.float(), b.tensors.float(), rtol=2e-2, atol=2e-2)


def test_eval_mode_returns_instances():
    model = small_model().eval()
    data = synthetic_batch(2, 224, 320, n_categories=4, boxes_per_image=3, device="cuda")
    with torch.no_grad():
        res = model(data)
    assert len(res) == 2
    inst = res[0]["instances"]
    assert inst.pred_boxes.tensor.shape == (20, 4) and inst.scores.shape == (20,)
    assert inst.pred_classes.dtype == torch.int64 and torch.isfinite(inst.pred_boxes.tensor).all()


def test_bf16_autocast_step_tracks_fp32():
    """BASELINE configs[3] runs the GEMMs in bf16 (autocast) around the fp32 native ops: the MSDA
    module and the RSB epilogue upcast their inputs, the decoder FFN stays in fp32 (reference
    transformer_for_adapter.py:1004), the criterion is fp32.  The loss dict must be finite, have
    the same keys and stay near the fp32 step (the side-branch losses, which sit in front of the
    top-k query selection, within bf16 rounding; the set losses within a factor of two -- with
    random-init logits near zero the selected queries change under bf16); gradients reach all 25
    side-branch tensors."""
    model = small_model().train()
    model.before_train()                      # freeze everything but the side branches
    model.use_frontend_graphs = False
    data = synthetic_batch(2, 224, 320, n_categories=4, boxes_per_image=3, device="cuda")
    ref = model(data)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(data)
    assert set(out) == set(ref)
    for k in ref:
        assert torch.isfinite(out[k]), k
        assert out[k].dtype == torch.float32, (k, out[k].dtype)
        if "adapter" in k:
            torch.testing.assert_close(out[k], ref[k], rtol=0.05, atol=1e-4, msg=k)
        else:
            assert 0.5 * float(ref[k]) - 0.05 <= float(out[k]) <= 2.0 * float(ref[k]) + 0.05, (k, float(out[k]), float(ref[k]))
    sum(out.values()).backward()
    grads = [(n, p.grad) for n, p in model.named_parameters() if p.requires_grad]
    assert len(grads) == 25
    assert all(g is not None and torch.isfinite(g).all() for _, g in grads)


def test_ragged_batch_different_sizes_and_captions():
    """Images of different sizes (padding masks at every level) and captions of different lengths
    (text padding) in one minibatch, one image without boxes: a training step runs, all losses
    are finite and eval produces per-image detections."""
    model = small_model().train()
    trainer = ZiraTrainer(model)
    a = synthetic_batch(1, 224, 320, n_categories=4, boxes_per_image=3, seed=1, device="cuda")[0]
    b = synthetic_batch(1, 200, 272, n_categories=2, boxes_per_image=1, seed=2, device="cuda")[0]
    from ziragroundingdino_amd.structures import Boxes, Instances
    inst = b["instances"]                                    # no ground truth in the second image
    b["instances"] = Instances(inst.image_size, gt_boxes=Boxes(inst.gt_boxes.tensor[:0]),
                               gt_classes=inst.gt_classes[:0])
    data = [a, b]
    assert a["captions"] != b["captions"]
    for _ in range(2):
        out = trainer.run_step(data)
        assert all(torch.isfinite(v) for v in out.values()), out
    model.eval()
    with torch.no_grad():
        res = model(data)
    assert len(res) == 2 and all("instances" in r for r in res)


def test_no_padding_shortcut_changes_nothing():
    """With equally sized images the key-padding fills are skipped (host-known ``no_padding``):
    the loss dict must be bit-identical to the run that applies the all-False masks."""
    model = small_model().train()
    model.before_train()
    model.use_frontend_graphs = False
    data = synthetic_batch(2, 224, 320, n_categories=4, boxes_per_image=3, device="cuda")
    from ziragroundingdino_amd import utils as zu

    fast = model(data)
    orig = zu.nested_tensor_from_tensor_list

    def padded(tl):
        out = orig(tl)
        out.no_padding = False
        return out

    import ziragroundingdino_amd.groundingdino as gd
    gd.nested_tensor_from_tensor_list = padded
    try:
        slow = model(data)
    finally:
        gd.nested_tensor_from_tensor_list = orig
    for k in slow:
        assert torch.equal(fast[k], slow[k]), k


def test_sampling_locations_one_pass_is_bit_identical():
    """ref + offsets / (W, H) as one addcdiv (float divisor) == the reference's two-op expression
    with its int64 divisor (ms_deform_attn.py:305-313): sampling pixels cannot move."""
    from ziragroundingdino_amd import ms_deform_attn as m
    g = torch.Generator().manual_seed(11)
    ref = torch.rand(2, 3000, 4, 2, generator=g).cuda()
    off = (torch.randn(2, 3000, 8, 4, 4, 2, generator=g) * 3).cuda()
    sh = torch.tensor([[100, 167], [50, 84], [25, 42], [13, 21]], device="cuda")
    try:
        m.FUSED_LOCATIONS = True
        a = m.sampling_locations_from_reference_points(ref, off, sh, 4)
        m.FUSED_LOCATIONS = False
        b = m.sampling_locations_from_reference_points(ref, off, sh, 4)
    finally:
        m.FUSED_LOCATIONS = True
    assert torch.equal(a, b)


@pytest.mark.parametrize("graph_encoder", [False, True])
def test_transformer_graph_path_matches_eager_and_survives_replays(graph_encoder, monkeypatch):
    """hipGraph replay of the decoder piece (the default) and of the encoder pieces as well (graphs.GraphedTransformer): same
    losses as the eager path step by step (stochastic depth off so that both see the same arithmetic), ten replays."""
    from ziragroundingdino_amd import graphs as zg
    monkeypatch.setattr(zg.GraphedTransformer, "graph_encoder", graph_encoder)

    def run(use_graph):
        model = small_model().train()            # seeds itself; stochastic depth is off in small_model
        model.use_transformer_graph = use_graph
        trainer = ZiraTrainer(model)
        data = synthetic_batch(2, 224, 320, n_categories=4, boxes_per_image=3, device="cuda")
        losses = []
        for _ in range(10):
            out = trainer.run_step(data)
            losses.append(float(sum(out.values())))
        torch.cuda.synchronize()
        return losses
    eager, graphed = run(False), run(True)
    assert all(torch.isfinite(torch.tensor(graphed)))
    for a, b in zip(eager, graphed):
        assert abs(a - b) <= 2e-3 * max(1.0, abs(a)), (eager, graphed)


_RCCL_ONE_RANK = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.environ["ZIRA_ROOT"], "tests")); sys.path.insert(0, os.environ["ZIRA_ROOT"])
from test_model_gpu import small_model
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
data = synthetic_batch(2, 224, 320, n_categories=4, boxes_per_image=3, device="cuda")
solo = ZiraTrainer(small_model().train())
solo.model.criterion.process_group = None
a = [solo.run_step(data) for _ in range(2)]
group = dist.new_group([0])
tr = ZiraTrainer(small_model().train(), process_group=group)
tr.always_reduce = True          # world size 1: issue the RCCL all-reduce of the flat bucket anyway
b = [tr.run_step(data) for _ in range(2)]
for x, y in zip(a, b):
    for k in x:
        torch.testing.assert_close(x[k], y[k], rtol=1e-4, atol=1e-5, msg=k)
# grad_value sums of sparse MSDA calls are not order-stable run to run (1e-7 relative); where the gradient of a
# zero-initialised side-branch weight is of the size of AdamW's eps (1e-8) that moves its update by a fraction of the
# learning rate: nearly all weights agree to 1e-5, the few others to a fraction of the two steps they have taken
for p, q in zip(solo.params, tr.params):
    d = (p - q).abs()
    assert float((d > 1e-5 + 1e-3 * q.abs()).float().mean()) < 0.01, float((d > 1e-5 + 1e-3 * q.abs()).float().mean())
    assert float(d.max()) < 0.2 * 2 * 1e-3, float(d.max())
dist.barrier(); torch.cuda.synchronize(); dist.destroy_process_group()
print("RCCL-ONE-RANK-OK")
"""


def test_one_rank_rccl_step_matches_no_collective():
    """The N>1 code path on the hardware at hand: a one-rank RCCL ("nccl" backend) process group, the
    criterion's num_boxes all-reduce and the flat side-branch bucket all-reduce issued on it, two steps;
    same losses and weights as the trainer without a process group.  (Own process: the group must not
    leak into the other tests.)"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300), ZIRA_ROOT=root,
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL-ONE-RANK-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


_TWO_RANKS_ONE_GPU = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.environ["ZIRA_ROOT"], "tests")); sys.path.insert(0, os.environ["ZIRA_ROOT"])
from test_model_gpu import small_model
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)                      # both ranks share the one GPU of the box
group = None
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)   # device tensors staged through the host
    group = dist.group.WORLD
model = small_model().train()
model.use_transformer_graph = True            # hipGraph replay of the transformer, as bench.py runs it
trainer = ZiraTrainer(model, process_group=group)
bucket = []
trainer.on_reduced_grad = lambda g: bucket.append(g.detach().cpu().clone()) if not bucket else None   # step 1, after the all-reduce
full = [synthetic_batch(2, 224, 320, n_categories=4, boxes_per_image=3, seed=s, device="cuda") for s in range(2)]
batches = [b[rank:rank + 1] for b in full] if world > 1 else full   # one image per rank / both images
losses = []
for it in range(4):                           # rotating minibatches, the next one's frozen front end prefetched
    out = trainer.run_step(batches[it % 2], next_data=batches[(it + 1) % 2])
    losses.append(float(sum(out.values())))
torch.cuda.synchronize()
assert trainer._prefetched is not None and all(l == l for l in losses)
if rank == 0:
    torch.save({"params": {n: p.detach().cpu() for n, p in zip(trainer.names, trainer.params)}, "losses": losses,
                "bucket": bucket[0]}, os.environ["ZIRA_OUT"])
if world > 1:
    dist.barrier(); dist.destroy_process_group()
print("TWO-RANKS-ONE-GPU-OK rank %d" % rank)
"""


def test_two_ranks_on_one_gpu_match_one_process(tmp_path):
    """The real N > 1 path on the hardware at hand: two fresh processes share cuda:0 over a gloo group (device
    tensors), each with the HIP kernels, the transformer replayed from hipGraphs, the front-end prefetch stream and
    the flat-bucket all-reduce live, one image per rank -- against ONE process stepping on both images (reference:
    DDP in train_multidatasets.py:406, num_boxes all-reduce in two_stage_criterion.py:59-65).  Four steps over two
    rotating minibatches."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = str(29900 + os.getpid() % 90)

    def launch(rank, world, out):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world),
                   ZIRA_ROOT=root, ZIRA_OUT=str(out), HSA_ENABLE_IPC_MODE_LEGACY="0")
        return subprocess.Popen([sys.executable, "-c", _TWO_RANKS_ONE_GPU], env=env, stdout=subprocess.PIPE,
                                stderr=subprocess.PIPE, text=True)

    procs = [launch(0, 2, tmp_path / "w2.pt"), launch(1, 2, tmp_path / "w2_r1.pt")]
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and "TWO-RANKS-ONE-GPU-OK" in so, so[-2000:] + se[-4000:]
    solo = launch(0, 1, tmp_path / "w1.pt")
    so, se = solo.communicate(timeout=900)
    assert solo.returncode == 0 and "TWO-RANKS-ONE-GPU-OK" in so, so[-2000:] + se[-4000:]
    w2, w1 = torch.load(tmp_path / "w2.pt"), torch.load(tmp_path / "w1.pt")
    # rank 0 reports the loss terms of ITS image (with the all-reduced normalisers); the weights must agree: sums over
    # images divided by the mean number of boxes, gradients averaged over the ranks.  (Tolerance as in _RCCL_ONE_RANK.)
    assert set(w1["params"]) == set(w2["params"])
    # the flat gradient bucket of step 1 after the all-reduce (mean over the two ranks) against the one-process gradient of
    # both images: the same sum in another order -- 1e-5 of the bucket's scale (measured: a few 1e-7)
    b2, b1 = w2["bucket"], w1["bucket"]
    assert b1.shape == b2.shape and torch.isfinite(b2).all()
    assert float((b2 - b1).abs().max()) <= 1e-5 * float(b1.abs().max()), (float((b2 - b1).abs().max()), float(b1.abs().max()))
    for n, q in w1["params"].items():
        p = w2["params"][n]
        assert torch.isfinite(p).all(), n
        d = (p - q).abs()
        assert float((d > 1e-5 + 1e-3 * q.abs()).float().mean()) < 0.01, (n, float((d > 1e-5 + 1e-3 * q.abs()).float().mean()))
        assert float(d.max()) < 0.2 * 4 * 1e-3, (n, float(d.max()))


def test_fused_query_projection_of_the_msda_module():
    """MultiScaleDeformableAttention with frozen weights projects the query ONCE for sampling offsets and attention
    weights (concatenated weights, one GEMM each way): same output and input gradients as the two projections; the
    cache follows in-place weight updates."""
    from ziragroundingdino_amd.ms_deform_attn import MultiScaleDeformableAttention as M

    torch.manual_seed(0)
    mod = M(embed_dim=256, num_heads=8, num_levels=3, num_points=4, batch_first=True).cuda()
    with torch.no_grad():
        mod.sampling_offsets.weight.normal_(0, 0.02)
        mod.attention_weights.weight.normal_(0, 0.1)
    for p in mod.parameters():
        p.requires_grad_(False)
    shapes = torch.tensor([[16, 20], [8, 10], [4, 5]], device="cuda")
    start = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    q = torch.randn(2, 50, 256, device="cuda", requires_grad=True)
    v = torch.randn(2, S, 256, device="cuda", requires_grad=True)
    ref = torch.rand(2, 50, 3, 2, device="cuda")
    g = torch.randn(2, 50, 256, device="cuda")

    def run():
        out = mod(query=q, value=v, reference_points=ref, spatial_shapes=shapes, level_start_index=start)
        return (out,) + torch.autograd.grad(out, [q, v], g)

    try:
        M.fuse_query_projections = False
        want = run()
        M.fuse_query_projections = True
        got = run()
        for a, b in zip(got, want):
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
        assert mod._fused_query_projection() is not None
        with torch.no_grad():
            mod.attention_weights.bias.add_(0.5)          # in-place update: the cached concatenation must follow
        M.fuse_query_projections = False
        want = run()
        M.fuse_query_projections = True
        got = run()
        for a, b in zip(got, want):
            torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)
    finally:
        M.fuse_query_projections = True


@pytest.mark.parametrize("R", [2, 4])
def test_sampling_plan_of_the_msda_module(R):
    """softmax + sampling locations in one native launch each way (csrc/sampling.hip) against the PyTorch op chain of the
    reference module (ms_deform_attn.py:290-325): the sampling locations bit for bit (the reference's roundings), the
    attention weights and the gradient of the projection within fp32 rounding; the module gives the same output and
    input gradients either way."""
    from ziragroundingdino_amd import ms_deform_attn as mda
    M = mda.MultiScaleDeformableAttention

    torch.manual_seed(1)
    mod = M(embed_dim=256, num_heads=8, num_levels=4, num_points=4, batch_first=True).cuda()
    with torch.no_grad():
        mod.sampling_offsets.weight.normal_(0, 0.05)
        mod.attention_weights.weight.normal_(0, 0.2)
    for p in mod.parameters():
        p.requires_grad_(False)
    shapes = torch.tensor([[16, 20], [8, 10], [4, 5], [2, 3]], device="cuda")
    start = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    q = torch.randn(2, 77, 256, device="cuda", requires_grad=True)
    v = torch.randn(2, S, 256, device="cuda", requires_grad=True)
    ref = torch.rand(2, 77, 4, R, device="cuda") * 0.6 + 0.2
    g = torch.randn(2, 77, 256, device="cuda")

    def run():
        value, loc, attn = mod.project(q, v, None, ref, shapes)
        out = mod(query=q, value=v, reference_points=ref, spatial_shapes=shapes, level_start_index=start)
        return (loc, attn, out) + torch.autograd.grad(out, [q, v], g)

    try:
        M.fuse_sampling_plan = False
        want = run()
        M.fuse_sampling_plan = True
        assert mda._sampling_plan_ok(torch.empty(1, 1, 384, device="cuda"), ref, shapes, 4, 4)
        got = run()
        assert torch.equal(got[0], want[0])                                   # sampling locations: the reference's roundings
        for a, b in zip(got[1:], want[1:]):
            torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-6)
        # the Function alone, with arbitrary incoming gradients
        proj = torch.randn(2, 77, 384, device="cuda", requires_grad=True)
        gl, ga = torch.randn(2, 77, 8, 4, 4, 2, device="cuda"), torch.randn(2, 77, 8, 4, 4, device="cuda")
        loc, attn = mda._SamplingPlan.apply(proj, ref, shapes, 8, 4, 4)
        gp = torch.autograd.grad([loc, attn], [proj], [gl, ga])[0]
        off, lg = proj.split([256, 128], dim=-1)
        loc_w = mda.sampling_locations_from_reference_points(ref, off.reshape(2, 77, 8, 4, 4, 2), shapes, 4)
        attn_w = lg.reshape(2, 77, 8, 16).softmax(-1).view(2, 77, 8, 4, 4)
        gp_w = torch.autograd.grad([loc_w, attn_w], [proj], [gl, ga])[0]
        assert torch.equal(loc, loc_w)
        torch.testing.assert_close(attn, attn_w, rtol=2e-6, atol=1e-7)
        torch.testing.assert_close(gp, gp_w, rtol=2e-5, atol=1e-6)
    finally:
        M.fuse_sampling_plan = True


def test_frontend_prefetch_gives_the_same_steps():
    """ZiraTrainer.run_step(data, next_data=...) queues the frozen front end of the next minibatch on a second stream
    and the next step picks it up: same losses and weights as without (no stochastic depth / dropout in this model)."""
    a_model, b_model = small_model().train(), small_model().train()
    a, b = ZiraTrainer(a_model), ZiraTrainer(b_model)
    batches = [synthetic_batch(2, 224, 320, n_categories=4, boxes_per_image=3, seed=s, device="cuda") for s in range(3)]
    seq = [batches[0], batches[1], batches[1], batches[2]]
    for i, data in enumerate(seq):
        la = a.run_step(data)
        nxt = seq[i + 1] if i + 1 < len(seq) else None
        lb = b.run_step(data, next_data=nxt)
        if nxt is not None:
            assert b._prefetched is not None and b._prefetched["inputs"] is nxt
        for k in la:
            torch.testing.assert_close(la[k], lb[k], rtol=1e-4, atol=1e-5, msg=k)
    for p, q in zip(a.params, b.params):   # (see the note in _RCCL_ONE_RANK)
        d = (p - q).abs()
        assert float((d > 1e-5 + 1e-3 * q.abs()).float().mean()) < 0.01 and float(d.max()) < 0.2 * 4 * 1e-3
    # a handle for another minibatch is ignored (the step computes its own front end)
    b._prefetched = b_model.prefetch_frontend(batches[0])
    lb = b.run_step(batches[2])
    assert all(torch.isfinite(v) for v in lb.values())


def test_add_layer_norm_equals_add_then_norm():
    """LayerNorm.add_norm(x, r) (one kernel: csrc/layernorm.hip with a residual input) against norm(x + r): the same
    forward bits, the same gradients for x, r (frozen affine: row kernel backward) and, with a trainable affine, for
    weight and bias (ATen's backward on the saved sum)."""
    from ziragroundingdino_amd.dense import LayerNorm

    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 5000, 256, generator=g).cuda().requires_grad_()
    r = torch.randn(2, 5000, 256, generator=g).cuda().requires_grad_()
    gy = torch.randn(2, 5000, 256, generator=g).cuda()
    for trainable in (False, True):
        ln = LayerNorm(256).cuda()
        with torch.no_grad():
            ln.weight.uniform_(0.5, 1.5)
            ln.bias.uniform_(-0.5, 0.5)
        ln.requires_grad_(trainable)
        want = ln(x + r)
        gw = torch.autograd.grad(want, [x, r] + ([ln.weight, ln.bias] if trainable else []), gy)
        got = ln.add_norm(x, r)
        gg = torch.autograd.grad(got, [x, r] + ([ln.weight, ln.bias] if trainable else []), gy)
        assert torch.equal(got, want)
        for a, b in zip(gg, gw):
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)
    small = torch.randn(4, 7, 256, generator=g).cuda()        # (below the row kernel's size: the plain path)
    assert torch.equal(LayerNorm(256).cuda().add_norm(small, small), LayerNorm(256).cuda()(small + small))


@pytest.mark.parametrize("graph_all", [False, True, "decoder_only"], ids=["default_pieces", "encoder_and_fusion_graphed", "decoder_only"])
def test_state_dict_loaded_after_capture_reaches_the_graphs(graph_all, monkeypatch):
    """A replayed hipGraph re-runs no Python: buffers derived from parameters (the MSDA modules' concatenated query
    projection, the decoder layers' transposed weights, the fusion blocks' composed text-side matrices) must follow a
    ``load_state_dict`` into a live model whose graphs exist already.  Two models, A stepped (graphs captured), then B's
    weights loaded into A: A's next loss must be the one B computes eagerly.  ``graph_all``: with the encoder pieces and the
    fusion blocks replayed too (supported switches; the composed text side then has no Python key check per step)."""
    from ziragroundingdino_amd.graphs import GraphedTransformer
    if graph_all == "decoder_only":              # (the default until round 6: encoder pieces launched eagerly)
        monkeypatch.setattr(GraphedTransformer, "graph_encoder", False)
    elif graph_all:
        monkeypatch.setattr(GraphedTransformer, "graph_encoder", True)
        monkeypatch.setattr(GraphedTransformer, "graph_fusion", True)
    a, b = small_model().train(), small_model().train()
    with torch.no_grad():
        for p in b.transformer.parameters():     # B: every transformer weight moved (incl. sampling_offsets / attention_weights)
            p.add_(0.05 * torch.randn_like(p))
    data = synthetic_batch(2, 224, 320, n_categories=4, boxes_per_image=3, device="cuda")
    a.use_transformer_graph, b.use_transformer_graph = True, False
    ta, tb = ZiraTrainer(a, lr=0.0), ZiraTrainer(b, lr=0.0)    # (lr 0: the steps leave the weights alone)
    before = float(sum(ta.run_step(data).values()))
    ta.run_step(data)
    a.load_state_dict(b.state_dict())
    got = float(sum(ta.run_step(data).values()))
    want = float(sum(tb.run_step(data).values()))
    assert abs(want - before) > 1e-2 * abs(want), "the two weight sets must give different losses for this test to bite"
    assert abs(got - want) <= 2e-3 * max(1.0, abs(want)), (got, want, before)


def test_product_graphs_never_capture_the_known_faulting_patterns(monkeypatch):
    """graphs.GraphedTransformer stays clear of what faulted on replay on ROCm 7.2 / torch 2.10 (scripts/repro_*.py):
    no ``torch.topk`` while a stream is capturing (query selection runs eagerly), no image <-> text fusion block inside a
    captured piece, and no memset node (``hipMemsetAsync`` is replayed out of order: the native library zero-fills with
    kernels).  One training step with graphs on; the guards fire inside the capture if the product ever regresses."""
    import ctypes

    from ziragroundingdino_amd import graphs as zg
    from ziragroundingdino_amd import transformer as zt

    seen = {"topk": 0, "fusion_in_capture": 0}
    real_topk = torch.topk

    def topk(*a, **k):
        if torch.cuda.is_current_stream_capturing():
            seen["topk"] += 1
        return real_topk(*a, **k)

    real_fwd = zt.BiAttentionBlock.forward

    def fusion_forward(self, *a, **k):
        if torch.cuda.is_current_stream_capturing():
            seen["fusion_in_capture"] += 1
        return real_fwd(self, *a, **k)

    monkeypatch.setattr(torch, "topk", topk)
    monkeypatch.setattr(zt.BiAttentionBlock, "forward", fusion_forward)
    assert zg.GraphedTransformer.graph_selection is False and zg.GraphedTransformer.graph_fusion is False
    model = small_model().train()
    model.use_transformer_graph = True
    trainer = ZiraTrainer(model)
    data = synthetic_batch(2, 224, 320, n_categories=4, boxes_per_image=3, device="cuda")
    for _ in range(3):
        out = trainer.run_step(data)
    torch.cuda.synchronize()
    assert all(torch.isfinite(v) for v in out.values())
    assert seen == {"topk": 0, "fusion_in_capture": 0}, seen
    # the native library itself: no hipMemsetAsync left in its import table
    import subprocess

    from ziragroundingdino_amd import _lib
    syms = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "hipMemsetAsync" not in syms and "hipMemset" not in syms, [l for l in syms.splitlines() if "Memset" in l]


_BENCH_UNDER_TORCHRUN = r"""
import json, os, subprocess, sys
root = os.environ["ZIRA_ROOT"]
cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
       "--master-port", os.environ["ZIRA_PORT"], os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
       "--no-cpu-baseline", "--no-second-mode", "--kernel-timing-steps", "1", "--height", "320", "--width", "448",
       "--force-collectives"]
p = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
line = json.loads(lines[0])
assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["parallelism"] == "dp1" and line["config"]["collectives_forced"]
assert line["roofline"]["inmodel_replay"]["dec"]["cold_pair_us"] > 0 and "micro" in line["roofline"]   # (the capture step is collective: all ranks)
print("BENCH-UNDER-TORCHRUN-OK %.2f images/s" % line["value"])
"""


def test_bench_under_torchrun_with_the_nccl_backend(tmp_path):
    """`bench.py` launched exactly as the driver launches the N > 1 runs (`python -m torch.distributed.run ... bench.py
    --gpus N`), with N = 1 -- the only N this box allows -- and `--force-collectives`: the nccl process group with
    `device_id=`, the barriers around the timed region, the MAX all-reduce of the timings, the flat-bucket all-reduce of
    every step and rank 0's one JSON line all execute on hardware (a small image keeps it short)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ZIRA_ROOT=root, ZIRA_PORT=str(29700 + os.getpid() % 90))
    p = subprocess.run([sys.executable, "-c", _BENCH_UNDER_TORCHRUN], env=env, capture_output=True, text=True, timeout=1700)
    assert p.returncode == 0 and "BENCH-UNDER-TORCHRUN-OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


_BENCH_TWO_RANKS = r"""
import json, os, subprocess, sys
root = os.environ["ZIRA_ROOT"]
cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
       "--master-port", os.environ["ZIRA_PORT"], os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
       "--no-second-mode", "--kernel-timing-steps", "1", "--height", "256", "--width", "320"]
env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", ZIRA_BENCH_DEVICE="0", ZIRA_BENCH_BACKEND="gloo")
p = subprocess.run(cmd, capture_output=True, text=True, timeout=1700, env=env)
lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-2000:], p.stderr[-4000:])
line = json.loads(lines[0])
assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["parallelism"] == "dp2" and line["config"]["global_batch"] == 4
assert line["scaling"] == "weak" and "cpu_baseline" not in line and line["roofline"]["inmodel_replay"]["dec"]["cold_pair_us"] > 0
print("BENCH-TWO-RANKS-OK %.2f images/s" % line["value"])
"""


def test_bench_control_flow_with_two_ranks_on_one_gpu():
    """The whole default `bench.py` path at N = 2 -- two ranks under `torch.distributed.run`, both pinned to the box's one
    GPU, collectives over gloo (RCCL refuses two ranks on a device): warm-up, the timed region between its barriers, the
    eager MSDA timing steps, the in-model capture step (collective: every rank), rank 0's micro-benchmarks while rank 1
    waits in the MAX all-reduce of the timings, one JSON line with the aggregate.  Round 4 found rank 0 running the capture
    step alone, which would have hung every N > 1 run in its first all-reduce."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ZIRA_ROOT=root, ZIRA_PORT=str(29500 + os.getpid() % 90))
    p = subprocess.run([sys.executable, "-c", _BENCH_TWO_RANKS], env=env, capture_output=True, text=True, timeout=1800)
    assert p.returncode == 0 and "BENCH-TWO-RANKS-OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def test_shared_source_and_multi_value_projections_change_nothing():
    """Round 4: the encoder's self-attention hands (src, pos) to its deformable-attention module as ONE source, so that
    value_proj(src) and the query projection of src + pos are one autograd node whose input gradients accumulate inside the
    second GEMM; the decoder projects the encoder output for all its layers' cross-attention in one node likewise
    (`multi_value_projections`).  Same outputs and gradients as the separate projections (summation order only)."""
    from ziragroundingdino_amd import transformer as zt
    from ziragroundingdino_amd.ms_deform_attn import MultiScaleDeformableAttention as M
    from ziragroundingdino_amd.ms_deform_attn import multi_value_projections

    torch.manual_seed(0)
    shapes = torch.tensor([[16, 20], [8, 10], [4, 5]], device="cuda")
    start = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    mods = [M(embed_dim=256, num_heads=8, num_levels=3, num_points=4, batch_first=True).cuda() for _ in range(3)]
    for mod in mods:
        with torch.no_grad():
            mod.sampling_offsets.weight.normal_(0, 0.02)
            mod.attention_weights.weight.normal_(0, 0.1)
            mod.value_proj.bias.normal_(0, 0.1)
        for p in mod.parameters():
            p.requires_grad_(False)
    src = torch.randn(2, S, 256, device="cuda", requires_grad=True)
    pos = torch.randn(2, S, 256, device="cuda")
    ref = torch.rand(2, S, 3, 2, device="cuda")
    mask = torch.zeros(2, S, dtype=torch.bool, device="cuda")
    mask[1, -37:] = True
    g = torch.randn(2, S, 256, device="cuda")

    def encoder_style(kpm):
        out = mods[0](query=src, query_pos=pos, value=src, reference_points=ref, spatial_shapes=shapes,
                      level_start_index=start, key_padding_mask=kpm)
        return (out,) + torch.autograd.grad(out, [src], g)

    for kpm in (None, mask):
        try:
            M.fuse_shared_source = False
            want = encoder_style(kpm)
        finally:
            M.fuse_shared_source = True
        got = encoder_style(kpm)
        for a, b in zip(got, want):
            torch.testing.assert_close(a, b, rtol=1e-4, atol=2e-5 * float(b.abs().max()))

    # decoder style: three modules read one memory, 40 queries each
    q = torch.randn(2, 40, 256, device="cuda", requires_grad=True)
    refq = torch.rand(2, 40, 3, 2, device="cuda")
    gq = torch.randn(2, 40, 256, device="cuda")

    def decoder_style(batched, kpm):
        vals = multi_value_projections(mods, src, kpm) if batched else None
        assert (vals is not None) == batched
        tot = 0
        for i, mod in enumerate(mods):
            tot = tot + mod(query=q, value=src, reference_points=refq, spatial_shapes=shapes, level_start_index=start,
                            key_padding_mask=kpm, value_projected=None if vals is None else vals[i])
        return (tot,) + torch.autograd.grad(tot, [q, src], gq)

    for kpm in (None, mask):
        want, got = decoder_style(False, kpm), decoder_style(True, kpm)
        for a, b in zip(got, want):
            torch.testing.assert_close(a, b, rtol=1e-4, atol=2e-5 * float(b.abs().max()))
    assert zt.TransformerDecoder.batch_value_projections is True


_RCCL_GROUP_RESTART = r"""
import itertools, os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.environ["ZIRA_ROOT"], "tests")); sys.path.insert(0, os.environ["ZIRA_ROOT"])
from test_model_gpu import small_model
from ziragroundingdino_amd.tasks import TaskSpec, run_tasks, multistep_lr_multiplier
from ziragroundingdino_amd.train import synthetic_batch
torch.cuda.set_device(0)
out = os.environ["ZIRA_OUT"]
datas = [synthetic_batch(2, 224, 320, n_categories=4, boxes_per_image=3, seed=s, device="cuda") for s in (0, 1)]
def spec(i, where):
    return TaskSpec(name="task%d" % i, categories_names=["fish%d" % i], data=lambda start, d=datas[i]: itertools.repeat(d), max_iter=3,
                    output_dir=os.path.join(out, where, "task%d" % i), lr_multiplier=multistep_lr_multiplier(2))
build = lambda: small_model().train()
def group(port):
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    return dist.group.WORLD
base = int(os.environ["ZIRA_PORT"])
# (a) both tasks under ONE process group
g = group(base)
whole = run_tasks([spec(0, "one"), spec(1, "one")], build, device="cuda", process_group=g)
dist.barrier(); torch.cuda.synchronize(); dist.destroy_process_group()
# (b) the group torn down and re-created between the tasks (a restarted communicator, as after a failed rank):
#     the second task starts from the first one's checkpoint under the NEW group
g = group(base + 1)
first = run_tasks([spec(0, "two")], build, device="cuda", process_group=g)
dist.barrier(); torch.cuda.synchronize(); dist.destroy_process_group()
assert not dist.is_initialized()
g = group(base + 2)
second = run_tasks([spec(1, "two")], build, init_checkpoint=first[0], device="cuda", process_group=g)
dist.barrier(); torch.cuda.synchronize(); dist.destroy_process_group()
a = torch.load(whole[1], weights_only=False)["model"]; b = torch.load(second[0], weights_only=False)["model"]
assert set(a) == set(b)
worst = 0.0
for n in a:
    if "adapter" in n:
        d = (a[n].float() - b[n].float()).abs()
        # (AdamW steps of lr-size on near-zero gradients: as in the one-rank test above)
        assert float((d > 1e-5 + 1e-3 * b[n].float().abs()).float().mean()) < 0.01, n
        worst = max(worst, float(d.max()))
assert worst < 0.2 * 6 * 1e-3, worst
print("RCCL-GROUP-RESTART-OK %.2e" % worst)
"""


def test_task_chain_survives_a_restart_of_the_rccl_group(tmp_path):
    """Two tasks of the chain (tasks.run_tasks: fresh model + previous ``model_final.pth``, the bucket all-reduce and the
    matching verdict on the process group) once under ONE nccl group and once with the group destroyed and re-created
    between the tasks -- three communicators in one process: the merged side-branch weights agree.  What an 8-GPU job
    does after losing a rank; here with world size 1, the only size this box allows."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", ZIRA_PORT=str(29300 + os.getpid() % 200), ZIRA_ROOT=root, ZIRA_OUT=str(tmp_path),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _RCCL_GROUP_RESTART], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "RCCL-GROUP-RESTART-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_front_end_queued_before_the_forward_completes_200_steps():
    """Regression test of the parked round-5 hang.  ROOT CAUSE (scripts/repro_streamk_two_streams.py, no model needed): every
    fp32 GEMM the library picks on gfx950 is a Stream-K kernel -- a persistent grid whose workgroups wait for their peers'
    partial sums --, and two streams that both run LARGE ones at the same time (the front end's Swin linears in their graph
    beside the encoder's 44 446-row products) leave each grid holding CU slots its peers need: the GPU never finishes.  With the
    package's own kernels for those products (the default arithmetic) the earlier placement runs; the trainer still queues the
    front end behind the encoder (train.py), where no large library GEMM of the step runs beside it in any arithmetic.
    Here: the earlier placement, default arithmetic, 200 steps, in a child process under a watchdog (a hang fails the test
    instead of the suite; the child is never re-executed)."""
    import os
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "repro_frontend_hang.py")
    out = subprocess.run([sys.executable, script, "at=start", "steps=200", "watchdog=60"], capture_output=True, text=True, timeout=400)
    assert out.returncode == 0 and "DONE" in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])
