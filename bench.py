#!/usr/bin/env python3
"""bench.py -- throughput of the hot path on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One JSON line on rank 0 (contract in the task statement).  A "step" is one forward + backward
pass of the hot path over one synthetic batch of `--batch` (default 2) 800x1333 images per GPU.

Workloads (``--workload``):
  msda_decoder   the MSDA op at the north-star shape B=2,Q=900,M=8,D=32,L=4,P=4,S=22223
                 (SURVEY.md section 8d): 1 fwd + 1 bwd per step, x `--calls` per step
                 (default 6 = the six decoder layers' cross-attention calls of one model step).

`roofline` is measured live: HIP events (on the stream the kernels are launched on -- torch's
current stream, which is what `_C` hands to the C ABI) around back-to-back launches of the
forward and of the backward entry point; the dominant one (backward) is reported.
`cpu_baseline` times the CPU oracle (oracle/msda_oracle.c, OpenMP over all host cores) on a
bounded number of repetitions of the same workload, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
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
    fwd = Vt + Lc + A + O
    bwd = O + Vt + 2 * Lc + 2 * A + V
    return fwd, bwd


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


def time_events(fn, iters, warmup=5):
    """Average device time of fn() over `iters` back-to-back launches on the current stream."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3  # seconds


def cpu_baseline_msda(B, Q, M, D, shapes, P, budget_s=12.0):
    """Oracle (kind 'port') on the host cores: fwd+bwd of the same workload, bounded sample."""
    from oracle import msda_oracle

    msda_oracle.build()
    value, sh, start, loc, attn, go = [t.cpu().numpy() for t in
                                       make_msda_inputs(B, Q, M, D, shapes, P, 0, "cpu")]
    cores = os.cpu_count() or 1
    msda_oracle.set_num_threads(cores)
    msda_oracle.msda_forward(value, sh, start, loc, attn)  # warm
    n, t0 = 0, time.perf_counter()
    while True:
        msda_oracle.msda_forward(value, sh, start, loc, attn)
        msda_oracle.msda_backward(go, value, sh, start, loc, attn)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 500:
            break
    return n, el, msda_oracle.num_threads()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="msda_decoder")
    ap.add_argument("--batch", type=int, default=2, help="images per GPU per step")
    ap.add_argument("--calls", type=int, default=6, help="MSDA fwd+bwd pairs per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from ziragroundingdino_amd import _C, _lib

    _lib.load()
    B, Q, M, D, P = args.batch, 900, 8, 32, 4
    shapes = NORTH_STAR_SHAPES
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    # each rank owns its own minibatch (data parallel, no data-path collective inside MSDA)
    value, sh, start, loc, attn, go = make_msda_inputs(B, Q, M, D, shapes, P, seed=rank, device=dev)

    def fwd():
        return _C.ms_deform_attn_forward(value, sh, start, loc, attn, 64)

    def bwd():
        return _C.ms_deform_attn_backward(value, sh, start, loc, attn, go, 64)

    def step():
        for _ in range(args.calls):
            fwd()
            bwd()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- roofline of the dominant kernel, measured live with events on the launch stream ----
    fwd_bytes, bwd_bytes = msda_algorithmic_bytes(B, S, M, D, L, Q, P)
    t_fwd = time_events(fwd, 200)
    t_bwd = time_events(bwd, 200)
    t_pair = time_events(lambda: (fwd(), bwd()), 200)

    if rank == 0:
        images = args.steps * B * world
        line = {
            "metric": "images/sec fwd+bwd GroundingDINO-T+ZiRa @800x1333 (hot-path workload: see config)",
            "value": images / elapsed,
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "%s: %d x (ms_deform_attn fwd+bwd) per step at B=%d,Q=%d,M=%d,D=%d,L=%d,P=%d,S=%d"
                            % (args.workload, args.calls, B, Q, M, D, L, P, S),
                "images_per_gpu": B,
                "parallelism": "dp%d" % world,
                "kernel_variant": _lib.variant_f32(D),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "ms_deform_attn backward (memset + msda_bwd kernel)",
                "achieved": bwd_bytes / t_bwd / 1e9,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": bwd_bytes / t_bwd / 1e9 / HBM_PEAK_GBS,
                "traffic": None,
                "algorithmic_bytes": bwd_bytes,
                "avg_us": t_bwd * 1e6,
                "fwd": {"achieved": fwd_bytes / t_fwd / 1e9, "frac": fwd_bytes / t_fwd / 1e9 / HBM_PEAK_GBS,
                        "algorithmic_bytes": fwd_bytes, "avg_us": t_fwd * 1e6},
                "fwd_bwd": {"achieved": (fwd_bytes + bwd_bytes) / t_pair / 1e9,
                            "frac": (fwd_bytes + bwd_bytes) / t_pair / 1e9 / HBM_PEAK_GBS,
                            "avg_us": t_pair * 1e6},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            n, el, cores = cpu_baseline_msda(B, Q, M, D, shapes, P)
            line["cpu_baseline"] = {
                "value": n * B / args.calls / el,
                "unit": "images/s",
                "cores": cores,
                "kind": "port",
                "sample": "%d x (oracle msda fwd+bwd, same shape, OpenMP %d threads) in %.1f s; "
                          "scaled by %d calls per step" % (n, cores, el, args.calls),
            }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
