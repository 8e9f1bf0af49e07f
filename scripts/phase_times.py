#!/usr/bin/env python3
"""Dev: GPU kernel time of a training step by phase, forward and backward separately.
Module forward hooks wrap the inputs / outputs of the big modules in an identity autograd Function that launches a
marker kernel (a fill with a distinctive element count) in forward and in backward; run under
  rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 scripts/phase_times.py run
then   python3 scripts/phase_times.py report DIR   splits the kernel trace at the markers."""
import csv, glob, os, re, sys, collections
MARK0 = 3_000_000          # marker fills have MARK0 + 4096 * id elements
NAMES = []

def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from ziragroundingdino_amd.config import zira_swint_config
    from ziragroundingdino_amd.groundingdino import build_model
    from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch
    from ziragroundingdino_amd import transformer as T
    dev = torch.device("cuda"); torch.manual_seed(0)
    model = build_model(zira_swint_config(device="cuda")).to(dev).train()
    model.use_transformer_graph = False   # eager launches: the hooks below sit inside the pieces a graph would hold
    trainer = ZiraTrainer(model)
    data = synthetic_batch(2, 800, 1333, device=dev)
    scratch = torch.empty(MARK0 + 4096 * 64, device=dev)
    ids = {}
    def mark(tag):
        i = ids.setdefault(tag, len(ids))
        scratch[: MARK0 + 4096 * i].fill_(0.0)
    class Mark(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, tag_f, tag_b):
            ctx.tag_b = tag_b
            mark(tag_f)
            return x.view_as(x)
        @staticmethod
        def backward(ctx, g):
            mark(ctx.tag_b)
            return g, None, None
    def first_tensor(obj):
        if torch.is_tensor(obj):
            return obj
        if isinstance(obj, (list, tuple)):
            for o in obj:
                t = first_tensor(o)
                if t is not None:
                    return t
        if isinstance(obj, dict):
            for o in obj.values():
                t = first_tensor(o)
                if t is not None:
                    return t
        return None
    def wrap(mod, name):
        # forward: marker "name" before the module, "after name" behind it; backward: the mirror image
        def pre(m, args, kwargs):
            args = list(args)
            done = False
            for k, a in enumerate(args):
                if torch.is_tensor(a) and a.is_floating_point():
                    args[k] = Mark.apply(a, "F " + name, "B end " + name) if a.requires_grad else (mark("F " + name), a)[1]
                    done = True
                    break
            if not done:
                for k, a in kwargs.items():
                    if torch.is_tensor(a) and a.is_floating_point():
                        kwargs[k] = Mark.apply(a, "F " + name, "B end " + name) if a.requires_grad else (mark("F " + name), a)[1]
                        done = True
                        break
            if not done:
                mark("F " + name)
            return tuple(args), kwargs
        def post(m, args, kwargs, out):
            def w(o):
                return Mark.apply(o, "F end " + name, "B " + name) if (torch.is_tensor(o) and o.is_floating_point() and o.requires_grad) else o
            if torch.is_tensor(out):
                if out.requires_grad:
                    return w(out)
                mark("F end " + name)
                return out
            if isinstance(out, tuple):
                res, done = [], False
                for o in out:
                    if not done and torch.is_tensor(o) and o.is_floating_point() and o.requires_grad:
                        res.append(w(o)); done = True
                    else:
                        res.append(o)
                if not done:
                    mark("F end " + name)
                return tuple(res)
            mark("F end " + name)
            return out
        mod.register_forward_pre_hook(pre, with_kwargs=True)
        mod.register_forward_hook(post, with_kwargs=True)
    tr = model.transformer
    for i, l in enumerate(tr.encoder.layers):
        wrap(l, "enc.deform")
    for i, l in enumerate(tr.encoder.text_layers):
        wrap(l, "enc.text")
    for i, l in enumerate(tr.encoder.fusion_layers):
        wrap(l, "enc.fusion")
    for i, l in enumerate(tr.decoder.layers):
        wrap(l, "dec.layer")
    wrap(tr.encoder, "ENCODER")
    wrap(tr.decoder, "DECODER")
    wrap(model.criterion, "criterion")
    for _ in range(3):
        trainer.run_step(data)
    torch.cuda.synchronize()
    mark("STEP")
    for _ in range(5):
        trainer.run_step(data)
        mark("STEP")
    torch.cuda.synchronize()
    with open("/tmp/phase_ids.txt", "w") as f:
        for k, v in ids.items():
            f.write("%d\t%s\n" % (v, k))

def report(d):
    ids = {}
    for l in open("/tmp/phase_ids.txt"):
        i, k = l.rstrip("\n").split("\t")
        ids[int(i)] = k
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # marker: FillFunctor<float> vectorized kernel whose grid covers MARK0 + 4096 i elements (4 per thread)
    def marker(r):
        if "FillFunctor<float>" not in r["Kernel_Name"]:
            return None
        g = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)))
        n = g * 4
        if n < MARK0 - 8192:
            return None
        i = round((n - MARK0) / 4096)
        return ids.get(i) if abs(n - (MARK0 + 4096 * i)) <= 2048 else None
    steps = 0
    cur = None
    acc = collections.defaultdict(lambda: [0.0, 0])
    stack = []
    started = False
    for r in rows:
        m = marker(r)
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if m is not None:
            if m == "STEP":
                steps += started
                started = True
                stack = []
                continue
            if not started:
                continue
            kind, rest = m.split(" ", 1)
            if rest.startswith("end "):
                name = rest[4:]
                while stack and stack[-1] != kind + " " + name:
                    stack.pop()
                if stack:
                    stack.pop()
            else:
                stack.append(kind + " " + rest)
            continue
        if not started or steps >= 5:
            continue
        key = " > ".join(stack) if stack else "(outside: backbone/bert graphs, heads, optimizer, input proj ...)"
        acc[key][0] += dur; acc[key][1] += 1
    steps = max(steps, 1)
    tot = sum(v[0] for v in acc.values())
    print("steps %d, kernel time per step %.2f ms" % (steps, tot / steps / 1e3))
    for k, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        print("%8.3f ms/step %7.1f launches/step  avg %6.2f us  %s" % (t / steps / 1e3, n / steps, t / n, k))

if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])
