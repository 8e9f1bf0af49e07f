#!/usr/bin/env python3
"""bench.py -- images/s of the ZiRa training step (GroundingDINO-T, 800x1333) on N MI355X GPUs.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One JSON line on rank 0.  A "step" is one pass of the hot path over one synthetic minibatch of
`--batch` (default 2) uint8 800x1333 images per GPU, i.e. BASELINE.json configs[1] per rank:
frozen Swin-T + BERT-base front end (random init -- no checkpoint offline), both ZiRa side
branches (fused RSB epilogue kernels), 6 encoder + 6 decoder layers with 12 MSDA forward and
12 backward calls on the gfx950 kernels, Hungarian matching + losses, backward, all-reduce of
the 4.6 M side-branch gradients (RCCL, N > 1), clip 0.1, AdamW -- nothing skipped.

`roofline`: every MSDA call of the timed steps is bracketed by HIP events on its launch stream
(`_C.TIMING`); calls are grouped into decoder shape (Q = 900, the north-star shape) and
encoder shape (Q = S), forward / backward, and priced with the algorithmic bytes of SURVEY.md
section 8(d).  The top-level object is the north-star kernel pair ("ms_deform_attn fwd+bwd at B=2,
Q=900, L=4, M=8, P=4"); `kernels` lists all four groups, `dominant` names the one with the
largest share of the step.
`roofline.micro`: the kernel micro-benchmark of SURVEY.md section 8(d), outside the timed region:
uniform / clustered sampling locations at the north-star shape and pixel-grid queries at the
encoder shape; 20 warm-up + 200 timed launches (50 at the encoder shape) of forward, backward and
the pair, replayed from a hipGraph so that the launches are back to back, timed with HIP events.
`cpu_baseline` (rank 0, N = 1): the same training step on the host cores -- this package's
model on CPU tensors with the two native MSDA entry points served by the CPU oracle (kind
"port"; the reference's own Python cannot travel to the GPU box) -- on a bounded sample; beside it
`cpu_baseline.msda`: the op alone on the section 8(d) inputs, as the reference's fallback computes
it (per-level F.grid_sample + autograd, the package's multi_scale_deformable_attn_pytorch, all host
cores) and as the C restatement (oracle/msda_oracle.c, OpenMP) does.

`python bench.py --gpus N` with N > 1 and no torchrun environment starts the N ranks itself (one
process per GPU through torch.distributed.run on 127.0.0.1) before anything touches the GPU.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
NORTH_STAR_SHAPES = [(100, 167), (50, 84), (25, 42), (13, 21)]


def msda_algorithmic_bytes(B, S, M, D, L, Q, P, esize=4):
    """unique bytes the op has to move, fwd and bwd."""
    V = B * S * M * D * esize
    Lc = B * Q * M * L * P * 2 * esize
    A = B * Q * M * L * P * esize
    O = B * Q * M * D * esize
    G = B * Q * M * L * P * 4 * D * esize
    Vt = min(V, G)
    return Vt + Lc + A + O, O + Vt + 2 * Lc + 2 * A + V


def make_msda_inputs(B, Q, M, D, shapes, P, seed, device):
    g = torch.Generator(device="cpu").manual_seed(seed)
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    value = torch.randn(B, S, M, D, generator=g)
    loc = torch.rand(B, Q, M, L, P, 2, generator=g)
    attn = torch.randn(B, Q, M, L * P, generator=g).softmax(-1).view(B, Q, M, L, P)
    grad_out = torch.randn(B, Q, M * D, generator=g)
    sh = torch.tensor(shapes, dtype=torch.long)
    start = torch.cat([sh.new_zeros(1), (sh[:, 0] * sh[:, 1]).cumsum(0)[:-1]])
    return [t.to(device) for t in (value, sh, start, loc, attn, grad_out)]


def graphed(fn, n):
    """Capture n back-to-back calls into a hipGraph so that replay is device-bound (the Python shim
    costs ~10 us of host time per call, more than the kernels take)."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    return g.replay


def timeit(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def msda_call_pair(_C, v, sh, st, loc, attn, go):
    """The forward and backward of one MSDA call as the autograd Function issues them: for sparse calls (decoder) the
    forward makes the backward's plan in its own launch and the backward starts from that plan; dense calls (encoder)
    have no plan.  Returns (fwd, bwd) closures; bwd uses the plan of the most recent fwd."""
    state = {}
    if _C.plan_applies(v, sh, st, loc, 64):
        def fwd():
            out, state["plan"] = _C.ms_deform_attn_forward_plan(v, sh, st, loc, attn, 64)
            return out
        fwd()
        bwd = lambda: _C.ms_deform_attn_backward(v, sh, st, loc, attn, go, 64, plan=state["plan"])
    else:
        fwd = lambda: _C.ms_deform_attn_forward(v, sh, st, loc, attn, 64)
        bwd = lambda: _C.ms_deform_attn_backward(v, sh, st, loc, attn, go, 64)
    return fwd, bwd


_FLUSH = {}


def timeit_cold(fn, iters, dev):
    """Average microseconds of ONE graph-replayed call behind a 512 MB write (twice the 256 MB Infinity Cache): HIP events
    bracket the replay of a one-call hipGraph, so the figure also holds the launch of that graph (~10 us) -- an upper bound,
    reported beside `timeit_cold_cycle`."""
    buf = _FLUSH.get(dev)
    if buf is None:
        buf = _FLUSH[dev] = torch.empty(128 * 1024 * 1024, dtype=torch.float32, device=dev)
    g = graphed(fn, 1)
    for _ in range(3):
        g()
    tot = 0.0
    for i in range(iters):
        buf.fill_(float(i))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g()
        e1.record()
        e1.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / iters * 1e3


def msda_cold_cycle(_C, v, sh, st, loc, attn, go, sets=8, reps=10):
    """Forward and backward of one MSDA call with COLD operands and no launch artefacts: `sets` independent copies of every
    operand (value alone is 45 MB; eight sets of inputs + outputs + plans are > 1 GB, four times the Infinity Cache), the
    calls of all sets captured back to back into one hipGraph per direction with every output kept alive (so that the
    allocator cannot hand a warm buffer to the next call); a call finds its operands where a training step leaves them --
    in HBM -- and the replayed graph has no host gaps.  Returns (fwd_us, bwd_us) per call."""
    copies = [tuple(t.clone() for t in (v, loc, attn, go)) for _ in range(sets)]
    planned = _C.plan_applies(v, sh, st, loc, 64)
    keep = []

    def fwd_all():
        keep.clear()
        for cv, cl, ca, _ in copies:
            if planned:
                keep.append(_C.ms_deform_attn_forward_plan(cv, sh, st, cl, ca, 64))
            else:
                keep.append((_C.ms_deform_attn_forward(cv, sh, st, cl, ca, 64), None))
    fwd_all()
    plans = [p for _, p in keep]
    outs = []

    def bwd_all():
        outs.clear()
        for (cv, cl, ca, cg), p in zip(copies, plans):
            kw = {} if p is None else {"plan": p}
            outs.append(_C.ms_deform_attn_backward(cv, sh, st, cl, ca, cg, 64, **kw))
    gb = graphed(bwd_all, 1)      # (uses the plans of the eager forward above; they stay alive in `plans`)
    tb = timeit(gb, reps) / sets
    outs.clear()
    gf = graphed(fwd_all, 1)
    tf = timeit(gf, reps) / sets
    return tf, tb


def encoder_loc(B, M, shapes, P, seed, dev):
    """Pixel-grid reference points of every level (reference transformer_for_adapter.py:482-497)
    + N(0, 2 px) offsets in each level's own pixels."""
    g = torch.Generator().manual_seed(seed)
    sh = torch.tensor(shapes, dtype=torch.float32)
    ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w,
                                                indexing="ij")[::-1], -1).reshape(-1, 2) for h, w in shapes])
    S, L = ref.shape[0], len(shapes)
    off = 2.0 * torch.randn(B, S, M, L, P, 2, generator=g)
    norm = torch.stack([sh[:, 1], sh[:, 0]], -1)[None, None, None, :, None, :]
    return (ref[None, :, None, None, None, :] + off / norm).contiguous().to(dev)


def clustered_loc(B, Q, M, L, P, seed, dev):
    """Decoder-like: box centres U(0.1, 0.9) + N(0, 0.05) offsets."""
    g = torch.Generator().manual_seed(seed)
    centre = torch.rand(B, Q, 1, 1, 1, 2, generator=g) * 0.8 + 0.1
    return (centre + 0.05 * torch.randn(B, Q, M, L, P, 2, generator=g)).to(dev)


def msda_micro(dev):
    """forward, backward and pair on synthetic inputs, device-bound."""
    from ziragroundingdino_amd import _C
    B, M, D, P, shapes = 2, 8, 32, 4, NORTH_STAR_SHAPES
    S, L = sum(h * w for h, w in shapes), len(shapes)
    v, sh, st, loc, attn, go = make_msda_inputs(B, 900, M, D, shapes, P, 0, dev)
    cases = [("northstar_uniform", (v, sh, st, loc, attn, go), 900, 200),
             ("northstar_clustered", (v, sh, st, clustered_loc(B, 900, M, L, P, 1, dev), attn, go), 900, 200)]
    captured = os.path.join(ROOT, "tests", "golden", "inmodel_decoder_locations.npz")
    if os.path.exists(captured):   # sampling locations / attention weights captured from a training step of the full-size model
        import numpy as np
        with np.load(captured) as z:
            cases.append(("northstar_inmodel", (v, sh, st, torch.from_numpy(z["loc"].astype(np.float32)).to(dev),
                                                torch.from_numpy(z["attn"].astype(np.float32)).to(dev), go), 900, 200))
    ve, _, _, _, attne, goe = make_msda_inputs(B, S, M, D, shapes, P, 2, dev)
    cases.append(("encoder_grid", (ve, sh, st, encoder_loc(B, M, shapes, P, 3, dev), attne, goe), S, 50))
    out = {}
    for name, (v, sh, st, loc, attn, go), Q, iters in cases:
        fb, bb = msda_algorithmic_bytes(B, S, M, D, L, Q, P)
        fwd, bwd = msda_call_pair(_C, v, sh, st, loc, attn, go)
        pair = lambda: (fwd(), bwd())
        per = 10
        res = {"Q": Q, "warmup": 20, "iters": iters}
        for key, fn, nbytes in (("fwd", fwd, fb), ("bwd", bwd, bb), ("pair", pair, fb + bb)):
            g = graphed(fn, per)
            timeit(g, 2)  # 20 warm-up launches
            us = timeit(g, max(1, iters // per)) / per
            res[key + "_us"] = us
            res[key + "_frac"] = nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS
        out[name] = res
    return out


def capture_inmodel(trainer, data):
    """One real (eager) step with a hook on the native backward: the MSDA inputs of the last decoder layer and the last
    encoder layer as the model produces them.  The step holds the data-parallel collectives: EVERY rank must call this."""
    from ziragroundingdino_amd import _C
    got = {}
    orig_b = _C.ms_deform_attn_backward

    def hook(value, sh, st, loc, attn, go, step, **kw):
        key = "enc" if loc.shape[1] == value.shape[1] else "dec"
        if key not in got:
            got[key] = [t.detach().clone() for t in (value, sh, st, loc, attn, go)]
        return orig_b(value, sh, st, loc, attn, go, step, **kw)

    use_graph = trainer.model.use_transformer_graph
    trainer.model.use_transformer_graph = False
    _C.ms_deform_attn_backward = hook
    try:
        trainer.run_step(data)
    finally:
        _C.ms_deform_attn_backward = orig_b
        trainer.model.use_transformer_graph = use_graph
    torch.cuda.synchronize()
    return got


def inmodel_replay(got, dev):
    """The MSDA calls of one real step (`capture_inmodel`), re-issued from hipGraphs: kernel time on the model's inputs
    without the launch gaps that HIP events around eager launches include -- back to back on one operand set (warm), and
    cycling through independent operand sets (cold).  No collectives: rank 0 alone runs this."""
    from ziragroundingdino_amd import _C
    torch.cuda.synchronize()
    out = {}
    for key, (v, sh, st, loc, attn, go) in got.items():
        B, S, M, D = v.shape
        Q, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
        fb, bb = msda_algorithmic_bytes(B, S, M, D, L, Q, P)
        fwd, bwd = msda_call_pair(_C, v, sh, st, loc, attn, go)
        per, iters = 10, (200 if key == "dec" else 50)
        res = {"dims_BSMDLQP": [B, S, M, D, L, Q, P], "warmup": 20, "iters": iters,
               "planned": bool(_C.plan_applies(v, sh, st, loc, 64))}
        for name, fn, nbytes in (("fwd", fwd, fb), ("bwd", bwd, bb), ("pair", lambda: (fwd(), bwd()), fb + bb)):
            g = graphed(fn, per)
            timeit(g, 2)
            us = timeit(g, max(1, iters // per)) / per
            res[name + "_us"] = us
            res[name + "_frac"] = nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS
        # the same two calls with cold operands: cycling through 8 independent operand sets inside one hipGraph per direction
        # (what a call costs inside the step), and one at a time behind a 512 MB write (includes the graph launch: upper bound)
        cyc_f, cyc_b = msda_cold_cycle(_C, v, sh, st, loc, attn, go, sets=8 if key == "dec" else 3)
        res.update({"cold_fwd_us": cyc_f, "cold_bwd_us": cyc_b, "cold_pair_us": cyc_f + cyc_b,
                    "cold_pair_frac": (fb + bb) / ((cyc_f + cyc_b) * 1e-6) / 1e9 / HBM_PEAK_GBS})
        if key == "dec":
            one_f, one_b = timeit_cold(fwd, 30, dev), timeit_cold(bwd, 30, dev)
            res.update({"cold_single_fwd_us": one_f, "cold_single_bwd_us": one_b, "cold_single_pair_us": one_f + one_b})
        out[key] = res
    return out


def cpu_baseline_msda(threads):
    """The op alone on the host cores, SURVEY.md section 8(d) inputs: the reference's fallback
    formulation (per-level grid_sample + autograd; ziragroundingdino_amd's product-side function)
    and the C restatement (test oracle), forward and forward+backward, bounded iteration counts."""
    import numpy as np
    from oracle import msda_oracle
    from ziragroundingdino_amd import multi_scale_deformable_attn_pytorch as msda_torch

    msda_oracle.build()
    msda_oracle.set_num_threads(threads)
    torch.set_num_threads(threads)
    B, M, D, P, shapes = 2, 8, 32, 4, NORTH_STAR_SHAPES
    S = sum(h * w for h, w in shapes)
    out = {}
    for name, Q, iters in (("northstar_B2_Q900", 900, 20), ("encoder_B2_Q22223", S, 5)):
        v, sh, st, loc, attn, go = make_msda_inputs(B, Q, M, D, shapes, P, 0, "cpu")
        if Q == S:
            loc = encoder_loc(B, M, shapes, P, 3, "cpu")
        vv, ll, aa = v.clone().requires_grad_(), loc.clone().requires_grad_(), attn.clone().requires_grad_()
        for _ in range(2):                       # warm-up (allocator, thread pools, first-call set-up)
            vv.grad = ll.grad = aa.grad = None
            msda_torch(vv, sh, ll, aa).backward(go)
        tf = tb = 0.0
        for _ in range(iters):
            vv.grad = ll.grad = aa.grad = None
            t0 = time.perf_counter()
            o = msda_torch(vv, sh, ll, aa)
            t1 = time.perf_counter()
            o.backward(go)
            t2 = time.perf_counter()
            tf += t1 - t0
            tb += t2 - t1
        npv = [x.numpy() for x in (v, sh, st, loc, attn, go)]
        msda_oracle.msda_forward(*npv[:5])
        msda_oracle.msda_backward(npv[5], *npv[:5])
        t0 = time.perf_counter()
        for _ in range(iters):
            msda_oracle.msda_forward(*npv[:5])
        t1 = time.perf_counter()
        for _ in range(iters):
            msda_oracle.msda_backward(npv[5], *npv[:5])
        t2 = time.perf_counter()
        out[name] = {"grid_sample_fwd_ms": tf / iters * 1e3, "grid_sample_fwd_bwd_ms": (tf + tb) / iters * 1e3,
                     "c_fwd_ms": (t1 - t0) / iters * 1e3, "c_fwd_bwd_ms": (t2 - t0) / iters * 1e3, "iters": iters}
    return out


def visible_gpus():
    """GPUs this process may use, counted WITHOUT touching the HIP runtime (on ROCm builds without the amdsmi
    path `torch.cuda.device_count()` falls through to hipGetDeviceCount, which brings the runtime up)."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    n = 0
    try:
        nodes = "/sys/class/kfd/kfd/topology/nodes"
        for d in os.listdir(nodes):
            with open(os.path.join(nodes, d, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            n += 1 if int(props.get("simd_count", "0")) > 0 else 0   # (CPU nodes have no SIMDs)
    except OSError:
        return None
    return n


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` outside torchrun: start N ranks (one per GPU) as a CHILD job and exit with
    its code.  The ranks must stay a child process (subprocess), never an exec of this one: on this pool a
    process that has initialised the GPU must not be replaced.  Nothing here touches the GPU."""
    import socket
    import subprocess
    have = visible_gpus()
    if have is not None and have < n:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible" % (n, have))
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def cpu_baseline_step(height=800, width=1333, sample_div=2, threads=None):
    """The same training step on the host cores, bounded: ONE step at batch 1 on an image whose
    sides are 1/`sample_div` of the benchmark's (1/4 of the pixels by default; a full-size step
    takes several minutes of CPU time), torch CPU ops for everything dense and the CPU oracle
    for the MSDA op.  Returns (equivalent full-size images, seconds, threads, description): the
    cost of the step is close to linear in the pixel count (Swin, the encoder's S tokens, the
    fusion S x T scores; only the 900-query decoder and BERT are fixed), so the sample counts
    as 1/sample_div^2 of a full-size image -- which flatters the CPU.
    """
    from oracle import msda_oracle
    from ziragroundingdino_amd import _C
    from ziragroundingdino_amd.config import zira_swint_config
    from ziragroundingdino_amd.groundingdino import build_model
    from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch

    msda_oracle.build()
    cores = os.cpu_count() or 1
    threads = threads or min(cores, 32)  # torch CPU ops stop scaling (and thrash) far below 256 threads
    torch.set_num_threads(threads)
    msda_oracle.set_num_threads(threads)
    saved = _C.ms_deform_attn_forward, _C.ms_deform_attn_backward
    # cpu_baseline leg only: the checker stands in for the kernels on CPU tensors
    _C.ms_deform_attn_forward = lambda v, s, st, l, a, step: torch.from_numpy(msda_oracle.msda_forward(
        v.detach().numpy(), s.numpy(), st.numpy(), l.detach().numpy(), a.detach().numpy()))
    _C.ms_deform_attn_backward = lambda v, s, st, l, a, go, step: [torch.from_numpy(x) for x in msda_oracle.msda_backward(
        go.detach().numpy(), v.detach().numpy(), s.numpy(), st.numpy(), l.detach().numpy(), a.detach().numpy())]
    try:
        torch.manual_seed(0)
        model = build_model(zira_swint_config(device="cpu")).train()
        trainer = ZiraTrainer(model)
        h, w = height // sample_div, width // sample_div
        data = synthetic_batch(1, h, w, device="cpu")
        trainer.run_step(data)                    # (first step: lazy set-up, allocator growth, thread pools -- not timed)
        t0 = time.perf_counter()
        trainer.run_step(data)
        el = time.perf_counter() - t0
        desc = ("the SECOND of 2 full training steps at batch 1 on a %dx%d image (1/%d of the %dx%d pixels; counted as "
                "that fraction of an image) on the host: this package's model on CPU tensors (torch CPU "
                "ops, %d of %d cores) with MSDA served by oracle/msda_oracle.c (OpenMP), %.1f s"
                % (h, w, sample_div * sample_div, height, width, threads, cores, el))
        return (h * w) / float(height * width), el, threads, desc
    finally:
        _C.ms_deform_attn_forward, _C.ms_deform_attn_backward = saved


def copy_ceiling(dev, nbytes=1 << 30, iters=10):
    """What a plain streaming kernel moves on THIS box: a 1 GiB device-to-device copy (bytes read +
    bytes written per second), a read-only pass (sum) and a write-only pass (fill) over the same 1 GiB, GB/s each."""
    n = nbytes // 4
    a = torch.empty(n, device=dev, dtype=torch.float32).normal_()
    b = torch.empty_like(a)

    def ev(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e-3

    t_copy, t_read, t_write = ev(lambda: b.copy_(a)), ev(lambda: a.sum()), ev(lambda: b.fill_(1.0))
    del a, b
    return {"bytes": nbytes, "copy_GBs": 2 * nbytes / t_copy / 1e9, "read_GBs": nbytes / t_read / 1e9,
            "write_GBs": nbytes / t_write / 1e9,
            "note": "1 GiB: torch copy_ (read + written bytes), sum (read only), fill_ (write only); HIP events, %d calls each" % iters}


def arithmetic_accuracy(dev):
    """The evidence for the configured arithmetic of the frozen products (reference: the FFN of transformer_for_adapter.py:877-886
    and the 256-wide projections of ms_deform_attn.py:262-288 under the freeze of groundingdino_dual_zero_rep_branch.py:722-745):
    on the encoder's own shape (44 446 rows), against an fp64 evaluation, the maximum and rms error -- relative to the largest
    |value| of the fp64 result -- of (a) the fused f16x2 FFN launches (csrc/ffn_f16x2.hip), forward and backward, beside the
    library's fp32 GEMM chain they replace, and (b) the split-bf16 product (csrc/gemm_bf16x3.hip) beside the library's fp32 GEMM
    on a 256-wide projection.  `ok` = no figure of the package exceeds the library's."""
    from ziragroundingdino_amd import ffn_f16x2 as ff, gemm_bf16x3 as g3
    M, F = 44446, 2048
    g = torch.Generator(device=dev).manual_seed(11)
    rn = lambda *shape: torch.randn(*shape, device=dev, generator=g)
    x, w1, b1, w2, b2 = rn(M, 256), rn(F, 256) * 0.06, rn(F) * 0.1, rn(256, F) * 0.03, rn(256) * 0.1
    gy, aux = rn(M, 256), rn(M, 256)
    pk = ff.PackedFFN()
    bits = ff.mask_like(x, F)
    ours_f = ff.run(x, pk.get(w1, b1, w2, False), F, False, bits, q_bias=b2)
    lib_h = torch._addmm_activation(b1, x, w1.t())
    lib_f = torch.addmm(b2, lib_h, w2.t())
    ours_b = ff.run(gy, pk.get(w1, b1, w2, True), F, True, bits, aux=aux)
    # the sign pattern the fused forward saved (a unit within rounding of zero may differ from fp64's): decoded from the bits
    words = bits.view(torch.int16).view(M, 2, F // 32).to(torch.int32) & 0xFFFF
    sign = torch.zeros(M, F, dtype=torch.bool, device=dev)
    for hf in range(2):
        for r in range(16):
            sign[:, 8 * (r // 4) + 4 * hf + r % 4::32] = ((words[:, hf, :] >> r) & 1).bool()
    lib_b = torch.addmm(aux, (gy @ w2) * sign, w1)
    acc = {k: [0.0, 0.0, 0.0] for k in ("ours_f", "lib_f", "ours_b", "lib_b")}   # max err, sum err^2, max |ref|
    n_el = 0
    for lo in range(0, M, 4096):
        sl = slice(lo, min(M, lo + 4096))
        ref_f = torch.addmm(b2.double(), torch.addmm(b1.double(), x[sl].double(), w1.double().t()).relu_(), w2.double().t())
        ref_b = aux[sl].double() + ((gy[sl].double() @ w2.double()) * sign[sl]) @ w1.double()
        for key, got, ref in (("ours_f", ours_f, ref_f), ("lib_f", lib_f, ref_f), ("ours_b", ours_b, ref_b), ("lib_b", lib_b, ref_b)):
            e = (got[sl].double() - ref).abs()
            acc[key][0] = max(acc[key][0], float(e.max()))
            acc[key][1] += float(e.pow(2).sum())
            acc[key][2] = max(acc[key][2], float(ref.abs().max()))
        n_el += ref_f.numel()
    rel = lambda k: {"max": acc[k][0] / acc[k][2], "rms": (acc[k][1] / n_el) ** 0.5 / acc[k][2]}
    out = {"shape": "M=%d rows, d_model 256, d_ffn %d, N(0,1) rows, weights N(0, 0.06 / 0.03)" % (M, F),
           "errors_are": "|result - fp64 result| / max |fp64 result|",
           "ffn_forward": {"f16x2_fused": rel("ours_f"), "library_fp32": rel("lib_f")},
           "ffn_backward": {"f16x2_fused": rel("ours_b"), "library_fp32": rel("lib_b")}}
    del ours_f, lib_h, lib_f, ours_b, lib_b, sign, words
    wp = rn(256, 256) * 0.06
    ref = x.double() @ wp.double().t()
    ours = g3.gemm(x, g3.split_planes(wp, False), g3.EPI_ADD, aux=torch.zeros(M, 256, device=dev)).double()
    lib = (x @ wp.t()).double()
    sc = float(ref.abs().max())
    out["projection_256"] = {"bf16x3": {"max": float((ours - ref).abs().max()) / sc, "rms": float((ours - ref).pow(2).mean().sqrt()) / sc},
                             "library_fp32": {"max": float((lib - ref).abs().max()) / sc, "rms": float((lib - ref).pow(2).mean().sqrt()) / sc}}
    pairs = [(out[k][a], out[k]["library_fp32"]) for k, a in (("ffn_forward", "f16x2_fused"), ("ffn_backward", "f16x2_fused"),
                                                                 ("projection_256", "bf16x3"))]
    out["ok"] = all(o["max"] <= l["max"] and o["rms"] <= l["rms"] for o, l in pairs)
    return out


def pmc_traffic():
    """HBM bytes per launch of the north-star kernel pair from the committed rocprofv3 PMC summary
    (profiles/*_msda_pmc_summary.json; FETCH_SIZE doubled as the gfx950 note in
    MI355X_MICROARCH.md prescribes for 16-B-per-lane loads, WRITE_SIZE as is).  PMC passes cannot
    run inside bench.py; the file is produced by scripts/collect_profiles.sh on the same shapes."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_msda_pmc_summary.json")))
    if not files:
        return None
    try:
        summary = json.load(open(files[-1]))
        dec = summary["decoder"]
        total = sum(k["hbm_read_bytes_corrected"] + k["hbm_write_bytes"] for k in dec.values())
        groups = {}   # the same per timed group: forward kernel / backward kernels of each shape
        for shape, tag in (("decoder", "dec"), ("encoder", "enc")):
            ks = summary.get(shape, {})
            fwd = [v for k, v in ks.items() if k.startswith("msda_fwd")]
            bwd = [v for k, v in ks.items() if k.startswith("msda_bwd")]
            if fwd:
                groups["fwd_" + tag] = sum(v["hbm_read_bytes_corrected"] + v["hbm_write_bytes"] for v in fwd)
            if bwd:
                groups["bwd_" + tag] = sum(v["hbm_read_bytes_corrected"] + v["hbm_write_bytes"] for v in bwd)
        return {"traffic": total, "traffic_source": os.path.relpath(files[-1], ROOT) +
                " (uniform sampling locations at the same shape; sum over " +
                ", ".join(sorted(k.split("<")[0] for k in dec)) + ")", "groups": groups}
    except Exception:
        return None


PROFILE_GROUPS = {   # kernels of a timed group as they are named in profiles/*_bench_kernel_stats.txt
    "fwd_dec": ("zira::msda_fwd_plan",),
    "bwd_dec": ("zira::msda_bwd_tile_accum<256u>", "zira::msda_bwd_fold"),   # (the fold launch: rounds 3-4 only)
    "fwd_enc": ("msda_fwd_lean<2>",),
    "bwd_enc": ("msda_bwd_bin", "msda_bwd_accum<32, 512>", "msda_bwd_fold<32>", "msda_bwd_walk<32, 4>"),
}


def profile_kernel_times():
    """Average microseconds per kernel name from the newest committed rocprofv3 summary of the bench command
    (profiles/*_bench_kernel_stats.txt, written by scripts/collect_profiles.sh), so that `roofline.kernels.*.frac_profiles`
    can be compared with the line's own event timing mechanically.  -> ({name: avg_us}, file) or ({}, None)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_kernel_stats.txt")))
    if not files:
        return {}, None
    avg, on = {}, False
    for ln in open(files[-1]):
        if ln.startswith("hand-written kernels"):
            on = True
            continue
        parts = ln.split(None, 4)
        if on and len(parts) == 5 and parts[0].endswith("%"):
            try:
                avg[parts[4].strip()] = float(parts[3])
            except ValueError:
                pass
    return avg, os.path.relpath(files[-1], ROOT)


def summarize_timing(records, B_expect):
    """Group the (kind, dims, e0, e1) records of `_C.TIMING` -> {group: (calls, avg seconds, dims)}."""
    groups = {}
    for kind, dims, e0, e1 in records:
        B, S, M, D, L, Q, P = dims
        shape = "enc" if Q == S else "dec"
        key = "%s_%s" % (kind, shape)
        ms = e0.elapsed_time(e1)
        g = groups.setdefault(key, [0, 0.0, dims])
        g[0] += 1
        g[1] += ms * 1e-3
    return {k: (v[0], v[1] / v[0], v[2]) for k, v in groups.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=2, help="images per GPU per step")
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1333)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-transformer-graph", dest="transformer_graph", action="store_false",
                    help="launch the transformer's kernels eagerly instead of replaying encoder / decoder layers from "
                         "hipGraphs (same kernels either way; eager, the step time follows the host's jitter)")
    ap.add_argument("--transformer-graph", dest="transformer_graph", action="store_true", help=argparse.SUPPRESS)
    ap.set_defaults(transformer_graph=True)
    ap.add_argument("--no-prefetch", dest="prefetch", action="store_false",
                    help="do not queue the frozen front end (Swin + BERT) of the next minibatch on a second stream "
                         "while the current step runs (same work per step either way)")
    ap.set_defaults(prefetch=True)
    ap.add_argument("--kernel-timing-steps", type=int, default=4,
                    help="eager steps after the timed region with HIP events around every MSDA call (roofline.kernels)")
    ap.add_argument("--cpu-sample-div", type=int, default=1,
                    help="cpu_baseline runs one step on an image with sides divided by this")
    ap.add_argument("--backbone", default="swin_T_224_1k",
                    help="swin_B_384_22k = BASELINE configs[3] (GroundingDINO-B); a separately labelled line")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="bf16: GEMMs under bf16 autocast around the fp32 native ops (configs[3])")
    ap.add_argument("--no-micro", action="store_true", help="skip the section 8(d) kernel micro-benchmark")
    ap.add_argument("--categories", type=int, default=15,
                    help="categories in the synthetic caption: 2 + 2 n text tokens (15 -> T = 32, the ODinW-like length of SURVEY.md 8d)")
    ap.add_argument("--minibatches", type=int, default=4,
                    help="distinct synthetic minibatches the steps rotate through (graph static buffers and the prefetch path see new data every step)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="with WORLD_SIZE=1 under torch.distributed.run: still create the nccl group and issue every barrier / "
                         "all-reduce of the N > 1 path (a hardware check of that path on a 1-GPU box)")
    ap.add_argument("--regions", type=int, default=3,
                    help="timed regions of --steps steps per launch mode; the line reports the MEDIAN region (each region is "
                         "bracketed by barrier + synchronize on both sides; all of them are listed in config.launch_modes)")
    ap.add_argument("--no-gemm-arith-mode", action="store_true",
                    help="skip the extra timed regions with the frozen products in the library's plain fp32 GEMMs (a child "
                         "process, reported beside the headline in config.gemm_arith)")
    ap.add_argument("--gemm-arith", default="f16x2", choices=["f32", "bf16x3", "f16x2"],
                    help="arithmetic of the frozen 44 446-row products for THIS process's timed steps.  f16x2 (the package's "
                         "default, transformer.Switches.gemm_arith): the encoder FFN as one fp32-accurate launch per direction "
                         "on the f16 matrix cores (csrc/ffn_f16x2.hip) and split-bf16 products for the other frozen "
                         "projections (csrc/gemm_bf16x3.hip); f32: the library's fp32 GEMMs.  The default run measures f32 in a "
                         "child process and reports it beside the headline, with the measured accuracy of both (`accuracy`)")
    ap.add_argument("--no-accuracy", action="store_true",
                    help="skip the accuracy block (fp64 comparison of the f16x2 / bf16x3 products and the library's fp32 GEMMs)")
    ap.add_argument("--no-second-mode", action="store_true",
                    help="skip the second timed region in the other launch mode (eager <-> hipGraph replay)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus, sys.argv[1:])  # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one process per GPU, each on the cores of its GPU's NUMA node -- set BEFORE the first GPU call (the eagerly launched
    # encoder pieces are host-sensitive, and eight unpinned ranks share and migrate across cores)
    from ziragroundingdino_amd import placement
    pinned = placement.pin_this_rank(verbose=(rank == 0 or os.environ.get("ZIRA_VERBOSE_PLACEMENT") == "1"))
    # The OTHER arithmetic (the library's plain fp32 GEMMs) is timed in a CHILD process of its own, run to completion BEFORE this process touches the GPU:
    # a second configuration timed in the process that has already captured and timed the first comes out 1-3 ms per step
    # slower whatever it is (scripts/ab_step.py's note; round 5: 36.5 ms in-process against 32.2 ms alone), and a process that
    # has initialised the GPU must not start programs on this pool.  One GPU, the flagship configuration, fp32 only.
    arith_child = None
    if (world == 1 and "WORLD_SIZE" not in os.environ and not args.no_gemm_arith_mode and args.dtype == "f32"
            and args.backbone == "swin_T_224_1k" and args.gemm_arith != "f32"):
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--gemm-arith", "f32", "--no-gemm-arith-mode", "--no-second-mode",
               "--no-micro", "--no-cpu-baseline", "--no-accuracy", "--kernel-timing-steps", "0", "--steps", str(args.steps),
               "--warmup", str(args.warmup), "--regions", str(args.regions), "--batch", str(args.batch), "--height", str(args.height),
               "--width", str(args.width), "--categories", str(args.categories), "--minibatches", str(args.minibatches)]
        cmd += [] if args.transformer_graph else ["--no-transformer-graph"]
        cmd += [] if args.prefetch else ["--no-prefetch"]
        try:
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            arith_child = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        except Exception as e:   # the headline does not depend on it
            print("[bench] plain-fp32 child run failed (%s); entry omitted" % (str(e).splitlines()[0] if str(e) else repr(e)),
                  file=sys.stderr, flush=True)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    # (test hooks: ZIRA_BENCH_DEVICE pins every rank to one GPU and ZIRA_BENCH_BACKEND=gloo carries the collectives through
    # the host, so that the N > 1 control flow of this file can run on a 1-GPU box -- RCCL refuses two ranks on one device)
    dev_index = int(os.environ.get("ZIRA_BENCH_DEVICE", local_rank))
    backend = os.environ.get("ZIRA_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist_on = world > 1 or (args.force_collectives and "WORLD_SIZE" in os.environ)
    if dist_on:
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from ziragroundingdino_amd import _C, _lib
    from ziragroundingdino_amd.config import zira_swint_config
    from ziragroundingdino_amd.groundingdino import build_model
    from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch

    _lib.load()  # fail loudly if the HIP extension is missing
    if dist_on:   # the launch mode is a flag, hence the same on every rank -- checked, because ranks in different modes would
        # still pass every collective and silently time different programs
        flag = torch.tensor([int(args.transformer_graph)], device=dev)
        lo, hi = flag.clone(), flag.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if int(lo) != int(hi):
            raise SystemExit("bench.py: ranks disagree on --transformer-graph")
    print("[bench] rank %d/%d on cuda:%d, launch mode %s, %s" % (
        rank, world, dev_index, "graph" if args.transformer_graph else "eager",
        "pinned to %d cores" % len(pinned) if pinned else "not pinned"), file=sys.stderr, flush=True)
    from ziragroundingdino_amd import transformer as _tr0
    arith_is_default = args.gemm_arith == _tr0.Switches.gemm_arith
    _tr0.Switches.gemm_arith = args.gemm_arith
    torch.manual_seed(0)  # identical replicas on every rank (as after loading one checkpoint)
    model = build_model(zira_swint_config(device=str(dev), backbone=args.backbone)).to(dev).train()
    model.use_transformer_graph = args.transformer_graph
    trainer = ZiraTrainer(model, amp_dtype=torch.bfloat16 if args.dtype == "bf16" else None,
                          process_group=dist.group.WORLD if dist_on else None)
    trainer.always_reduce = dist_on and world == 1   # (--force-collectives: the bucket all-reduce on a one-rank group too)
    # own shard: `--minibatches` distinct minibatches per rank, visited in turn
    batches = [synthetic_batch(args.batch, args.height, args.width, n_categories=args.categories,
                               seed=rank * 1009 + i, device=dev) for i in range(max(1, args.minibatches))]
    data = batches[0]
    cursor = [0]

    def run_steps(n):
        for _ in range(n):
            cur = batches[cursor[0] % len(batches)]
            nxt = batches[(cursor[0] + 1) % len(batches)]
            cursor[0] += 1
            trainer.run_step(cur, next_data=nxt if args.prefetch else None)

    def timed(n):
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(n)
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    try:
        run_steps(args.warmup)
    except RuntimeError as e:
        # Only a refused graph capture is retried eagerly: anything else (out of memory, a launch error, a device
        # assert) ends the run -- and so does a capture that was left open (the stream would still be capturing).
        msg = str(e).splitlines()[0] if str(e) else repr(e)
        capture = any(w in str(e).lower() for w in ("captur", "graph"))
        if not args.transformer_graph or not capture or torch.cuda.is_current_stream_capturing():
            raise
        print("[bench] transformer graph capture failed (%s); falling back to eager launches" % msg,
              file=sys.stderr, flush=True)
        args.transformer_graph = False
        model.use_transformer_graph = False
        trainer.flat_grad.zero_()
        run_steps(args.warmup)
    regions = max(1, args.regions)
    _C.TIMING = []
    all_regions = {"graph" if args.transformer_graph else "eager": [timed(args.steps) for _ in range(regions)]}
    records, _C.TIMING = _C.TIMING, None
    if not args.no_second_mode and args.dtype == "f32":   # the same steps in the other launch mode (every rank: collectives)
        other = not args.transformer_graph
        model.use_transformer_graph = other
        try:
            run_steps(max(2, args.warmup))
            all_regions["graph" if other else "eager"] = [timed(args.steps) for _ in range(regions)]
        except RuntimeError as e:
            if not other or torch.cuda.is_current_stream_capturing():
                raise
            print("[bench] second mode skipped (%s)" % (str(e).splitlines()[0] if str(e) else repr(e)), file=sys.stderr, flush=True)
        model.use_transformer_graph = args.transformer_graph
    # The same steps with the frozen products in the library's plain fp32 GEMMs, reported BESIDE the headline (config.gemm_arith).
    arith_regions = arith_how = None
    if arith_child is not None:
        try:
            mode = arith_child["config"]["launch_modes"][arith_child["config"]["launch_mode"]]
            arith_regions = [x * args.steps / 1e3 for x in mode["regions_ms_per_step"]]
            arith_how = ("a child process of this command (python bench.py --gemm-arith f32 ...), run before this process "
                         "touched the GPU; same GPU, launch mode %s" % arith_child["config"]["launch_mode"])
        except (KeyError, TypeError):
            pass
    timing_source = "HIP events around every native MSDA call of the timed steps"
    if args.transformer_graph and args.kernel_timing_steps > 0:   # (every rank: the steps hold collectives)
        # graph replays hide the launches from event timing: the same step, launched eagerly, right after
        model.use_transformer_graph = False
        trainer.run_step(data)          # (no prefetch here: a concurrent Swin would sit inside the event brackets)
        torch.cuda.synchronize()
        _C.TIMING = []
        for _ in range(args.kernel_timing_steps):
            trainer.run_step(data)
        torch.cuda.synchronize()
        records, _C.TIMING = _C.TIMING, None
        model.use_transformer_graph = True
        timing_source = ("HIP events around every native MSDA call of %d eager steps run right after the timed "
                         "region (its steps replay the transformer from hipGraphs, which hides the launches)"
                         % args.kernel_timing_steps)
    replay = None
    if not args.no_micro:
        got = capture_inmodel(trainer, data)     # (a training step: every rank, the collectives inside need them all)
        if rank == 0:
            replay = inmodel_replay(got, dev)
        del got
    if dist_on and arith_regions is not None:
        t = torch.tensor(arith_regions, device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        arith_regions = [float(x) for x in t.tolist()]
    if dist_on:   # every region: the MAX over ranks
        keys = sorted(all_regions)
        t = torch.tensor([x for k in keys for x in all_regions[k]], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        flat = [float(x) for x in t.tolist()]
        all_regions = {k: flat[i * regions:(i + 1) * regions] for i, k in enumerate(keys)}
    # Both launch modes ran the SAME K steps, `--regions` times each, every region between the same barriers (max over
    # ranks).  `value` is the CONFIGURED mode (the trainer's default unless a flag says otherwise), its median region; the
    # other mode stands beside it in `config.launch_modes` and is never the headline.
    modes = {k: sorted(v)[len(v) // 2] for k, v in all_regions.items()}
    primary_mode = "graph" if args.transformer_graph else "eager"
    reported_mode = primary_mode
    elapsed = modes[reported_mode]

    peak_mem = int(torch.cuda.max_memory_allocated(dev))          # (of the timed steps: read before the side measurements below)
    rank_devices = [[rank, dev_index, torch.cuda.get_device_name(dev)]]
    if dist_on:
        gathered = [None] * dist.get_world_size()
        dist.all_gather_object(gathered, rank_devices[0])
        rank_devices = gathered
        t = torch.tensor([peak_mem], device=dev, dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        peak_mem = int(t)

    if rank == 0:
        groups = summarize_timing(records, args.batch)
        timed_steps = args.kernel_timing_steps if (args.transformer_graph and args.kernel_timing_steps > 0) else args.steps
        kernels = {}
        for key, (calls, avg_s, dims) in sorted(groups.items()):
            fb, bb = msda_algorithmic_bytes(*dims)
            nbytes = fb if key.startswith("fwd") else (bb if key.startswith("bwd") else 0)
            kernels[key] = {"calls_per_step": calls / timed_steps, "avg_us": avg_s * 1e6,
                            "algorithmic_bytes": nbytes, "achieved": nbytes / avg_s / 1e9,
                            "frac": nbytes / avg_s / 1e9 / HBM_PEAK_GBS,
                            "share_of_step": (calls / timed_steps) * avg_s / (elapsed / args.steps),
                            "dims_BSMDLQP": list(dims)}
        prof_avg, prof_file = profile_kernel_times()
        for key, names in PROFILE_GROUPS.items():   # the same groups from the committed rocprofv3 summary of this command
            if key in kernels and all(n in prof_avg for n in names[:1]):
                us = sum(prof_avg.get(n, 0.0) for n in names)
                kernels[key]["avg_us_profiles"] = us
                kernels[key]["frac_profiles"] = kernels[key]["algorithmic_bytes"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS
                kernels[key]["profiles_source"] = prof_file + ": " + " + ".join(n for n in names if n in prof_avg)
        dominant = max(kernels, key=lambda k: kernels[k]["share_of_step"]) if kernels else None
        roofline = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
                    "kernels": kernels, "dominant": dominant, "timing_source": timing_source}
        pmc = pmc_traffic()
        if pmc:
            for key, nbytes in pmc.pop("groups", {}).items():   # HBM bytes per launch of every group (PMC, uniform / grid inputs)
                if key in kernels:
                    kernels[key]["traffic"] = nbytes
            roofline.update(pmc)
        if "fwd_dec" in kernels and "bwd_dec" in kernels:
            f, b = kernels["fwd_dec"], kernels["bwd_dec"]
            t_pair = (f["avg_us"] + b["avg_us"] + (kernels["plan_dec"]["avg_us"] if "plan_dec" in kernels else 0.0)) * 1e-6
            nbytes = f["algorithmic_bytes"] + b["algorithmic_bytes"]
            roofline.update({"kernel": "ms_deform_attn fwd+bwd, decoder cross-attention shape "
                                       "B=%d,S=%d,M=%d,D=%d,L=%d,Q=%d,P=%d" % tuple(f["dims_BSMDLQP"]),
                             "achieved": nbytes / t_pair / 1e9, "frac": nbytes / t_pair / 1e9 / HBM_PEAK_GBS,
                             "algorithmic_bytes": nbytes, "avg_us": t_pair * 1e6})
        if not args.no_micro:
            roofline["micro"] = msda_micro(dev)
        if replay:
            roofline["inmodel_replay"] = replay
            if "dec" in replay:
                # The headline figure: the model's own decoder call, forward (+ plan) and backward, replayed from hipGraphs
                # that cycle through eight independent operand sets (> 1 GB: every call finds its operands in HBM, as in
                # the step).  The warm back-to-back replay of round 3's headline and the eager event brackets of the real
                # step (host launch gaps included) stay beside it.
                r = replay["dec"]
                fb, bb = msda_algorithmic_bytes(*r["dims_BSMDLQP"])
                roofline["eager_events"] = {k: roofline.get(k) for k in ("achieved", "frac", "avg_us")}
                roofline["warm_replay"] = {"achieved": (fb + bb) / (r["pair_us"] * 1e-6) / 1e9, "frac": r["pair_frac"],
                                           "avg_us": r["pair_us"]}
                roofline.update({"achieved": (fb + bb) / (r["cold_pair_us"] * 1e-6) / 1e9, "frac": r["cold_pair_frac"],
                                 "avg_us": r["cold_pair_us"], "algorithmic_bytes": fb + bb,
                                 "frac_source": "inmodel_replay.dec.cold_pair_us: the decoder MSDA call captured from a "
                                                "training step, forward (with the backward's plan) and backward replayed "
                                                "from hipGraphs cycling through 8 independent operand sets (cold operands, "
                                                "no host gaps); cold_single_* = one call behind a 512 MB write, graph launch "
                                                "included; "
                                                "warm_replay = the same calls back to back (round 3's headline); "
                                                "eager_events = " + timing_source})
        # what a plain streaming kernel moves on this box, and the headline against it
        if not args.no_micro:
            try:
                roofline["copy_ceiling"] = copy_ceiling(dev)
                if roofline.get("achieved"):
                    roofline["frac_of_copy_ceiling"] = roofline["achieved"] / roofline["copy_ceiling"]["copy_GBs"]
            except RuntimeError as e:
                print("[bench] copy ceiling skipped (%s)" % str(e).splitlines()[0], file=sys.stderr, flush=True)
        images = args.steps * args.batch * world
        flagship = args.backbone == "swin_T_224_1k" and args.dtype == "f32"
        accuracy = None
        if flagship and world == 1 and not args.no_accuracy and args.gemm_arith != "f32":
            try:
                accuracy = arithmetic_accuracy(dev)
            except RuntimeError as e:
                print("[bench] accuracy block skipped (%s)" % str(e).splitlines()[0], file=sys.stderr, flush=True)
        # scratch the native paths hold: the decoder's six backward plans (forward -> backward), the dense MSDA backward's
        # workspace (one, reused), the fused FFN's shares (one per stream)
        lib_ = _lib.load()
        S_tok = sum(-(-args.height // d) * -(-args.width // d) for d in (8, 16, 32, 64))
        mem = {"peak_allocated_bytes": peak_mem,
               "msda_plan_bytes_per_decoder_layer": int(lib_.zira_msda_plan_bytes(args.batch, S_tok, 8, 32, 4, 900, 4)),
               "msda_dense_backward_workspace_bytes": int(lib_.zira_msda_bwd_workspace_bytes(args.batch, S_tok, 8, 32, 4, S_tok, 4)),
               "ffn_f16x2_workspace_bytes": int(lib_.zira_ffn_f16x2_workspace_bytes(args.batch * S_tok, 2048)),
               "note": "peak_allocated_bytes = torch.cuda.max_memory_allocated over the timed steps (max over ranks)"}
        size = "T" if args.backbone.startswith("swin_T") else "B"
        line = {
            "metric": "images/sec fwd+bwd GroundingDINO-%s+ZiRa @800x1333" % size,
            "value": images / elapsed,
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": ("configs[1]: GroundingDINO-T (Swin-T + BERT-base, random init) frozen-backbone "
                             if flagship else
                             "configs[3] variant (%s, %s GEMMs; NOT the headline config): GroundingDINO-%s frozen-backbone "
                             % (args.backbone, args.dtype, size)) +
                            "ZiRa side-branch fine-tune step, %d x %dx%d images per GPU, 900 queries, "
                            "full fwd+loss+bwd+clip+AdamW" % (args.batch, args.height, args.width),
                "backbone": args.backbone,
                "images_per_gpu": args.batch,
                "global_batch": args.batch * world,
                "parallelism": "dp%d" % world, "transformer_graph": bool(args.transformer_graph),
                "launch_mode": reported_mode, "launch_mode_default": primary_mode,
                "graphed_pieces": (lambda G: {"encoder_layers": bool(G.graph_encoder), "decoder": bool(G.graph_decoder),
                                              "fusion_blocks": bool(G.graph_fusion), "query_selection": bool(G.graph_selection),
                                              "front_end": True})(__import__("ziragroundingdino_amd.graphs", fromlist=["GraphedTransformer"]).GraphedTransformer)
                if args.transformer_graph else {},
                "frontend_prefetch": bool(args.prefetch), "collectives_forced": bool(dist_on and world == 1),
                "text_tokens": 2 + 2 * args.categories, "distinct_minibatches": len(batches),
                "launch_modes": {k: {"images_per_s": images / v, "ms_per_step": v / args.steps * 1e3,
                                     "regions_ms_per_step": [x / args.steps * 1e3 for x in all_regions[k]]}
                                 for k, v in sorted(modes.items())},
                "value_is": "median of %d timed regions of %d steps in the configured launch mode (%s)"
                            % (regions, args.steps, primary_mode)
                            + ("; arithmetic of the frozen products: %s (%s)" % (
                                args.gemm_arith, "the package's configured default, transformer.Switches.gemm_arith"
                                if arith_is_default else "NOT the package default: started with --gemm-arith")),
                "gemm_arith_configured": args.gemm_arith,
                "gemm_arith": {args.gemm_arith: {"images_per_s": images / elapsed, "ms_per_step": elapsed / args.steps * 1e3,
                                                 "note": {"f32": "the library's fp32 GEMMs for every frozen product",
                                                          "bf16x3": "split-bf16 products (csrc/gemm_bf16x3.hip) for the frozen "
                                                                    "44 446-row products, fp32-accurate",
                                                          "f16x2": "`value`: the encoder FFN as one fp32-accurate launch per "
                                                                   "direction on the f16 matrix cores (csrc/ffn_f16x2.hip), "
                                                                   "split-bf16 products (csrc/gemm_bf16x3.hip) for the other "
                                                                   "frozen projections; outputs stay fp32 and closer to fp64 "
                                                                   "than the library's (see `accuracy`)"}[args.gemm_arith]},
                               **({"f32": {"images_per_s": images / sorted(arith_regions)[len(arith_regions) // 2],
                                           "ms_per_step": sorted(arith_regions)[len(arith_regions) // 2] / args.steps * 1e3,
                                           "regions_ms_per_step": [x / args.steps * 1e3 for x in arith_regions],
                                           "note": "the same steps with the library's plain fp32 GEMMs for every frozen product "
                                                   "(last round's headline arithmetic); same launch mode as `value`",
                                           "measured_in": arith_how}}
                                  if arith_regions else {})},
                "rccl_ranks": (dist.get_world_size() if dist_on else 1), "rank_devices": rank_devices,
                "trainable_values": int(trainer.flat_grad.numel()),
                "ranks_pinned_to_numa_cores": bool(pinned),
                "msda_kernel_variant": _lib.variant_f32(32),
            },
            "roofline": roofline,
            "memory": mem,
        }
        if accuracy is not None:
            line["accuracy"] = accuracy
        if world == 1 and not args.no_cpu_baseline and flagship:
            n_img, el, cores, desc = cpu_baseline_step(args.height, args.width, args.cpu_sample_div)
            line["cpu_baseline"] = {"value": n_img / el, "unit": "images/s", "cores": cores,
                                    "cores_on_host": os.cpu_count(), "kind": "port", "sample": desc, "msda": cpu_baseline_msda(cores)}
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
