#!/usr/bin/env python3
"""Dev: which LINES of this package issue the ATen launches of one eager training step at the benchmark size.  A
TorchDispatchMode logs every ATen op that runs a kernel (views and metadata ops are skipped) with the innermost frame of this
package on the Python stack; ops that the autograd engine runs (no Python frame) are attributed to the line that built their
node (anomaly mode records it) and marked "bwd".  `python scripts/launch_sources.py [substring of op name]`;
`BIG=1`: only ops whose first tensor has >= 1e6 elements."""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ziragroundingdino_amd.config import zira_swint_config  # noqa: E402
from ziragroundingdino_amd.groundingdino import build_model  # noqa: E402
from ziragroundingdino_amd.train import ZiraTrainer, synthetic_batch  # noqa: E402

want = sys.argv[1] if len(sys.argv) > 1 else ""
big_only = os.environ.get("BIG") == "1"
VIEWS = ("view", "reshape", "transpose", "expand", "slice", "select", "permute", "unsqueeze", "squeeze", "as_strided", "detach",
         "alias", "_unsafe_view", "unbind", "split", "t.default", "unflatten", "flatten", "chunk", "narrow", "size", "stride",
         "is_", "sym_", "_local_scalar", "item", "empty", "new_empty", "result_type", "lift", "numel", "dim", "set_", "resize_",
         "_reshape_alias", "unfold", "movedim", "swapaxes", "view_as", "diagonal", "record_stream", "_to_copy.default:meta", "prim")
PKG = os.sep + "ziragroundingdino_amd" + os.sep


def pkg_frame_from_stack():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if PKG in fr.filename:
            return "%s:%d %s" % (fr.filename.split(PKG)[-1], fr.lineno, fr.name)
    return None


def pkg_frame_from_node(node):
    tb = node.metadata.get("traceback_") if node is not None else None
    if not tb:
        return None
    for line in reversed(tb):
        if PKG in line and 'File "' in line:
            f = line.split('File "')[1]
            path, rest = f.split('", line ')
            ln = rest.split(",")[0]
            fn = rest.split(" in ")[-1].split("\n")[0]
            return "%s:%s %s" % (path.split(PKG)[-1], ln, fn)
    return None


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()
        self.total = 0

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.__name__ if hasattr(func, "__name__") else str(func)
        if any(name.startswith(v) for v in VIEWS) or (want and want not in name):
            return out
        shape = None
        for a in list(args) + [out]:
            if isinstance(a, torch.Tensor):
                if a.is_cuda:
                    shape = tuple(a.shape)
                break
            if isinstance(a, (list, tuple)) and a and isinstance(a[0], torch.Tensor):
                shape = ("list",) + tuple(a[0].shape)
                break
        if shape is None:
            return out
        if big_only:
            n = 1
            for d in shape:
                n *= d if isinstance(d, int) else 1
            if n < 1_000_000:
                return out
        src = pkg_frame_from_stack()
        if src is None:
            node = torch._C._current_autograd_node() if hasattr(torch._C, "_current_autograd_node") else None
            src = pkg_frame_from_node(node)
            src = ("bwd of %s @ %s" % (type(node).__name__ if node is not None else "?", src)) if node is not None else "(no frame)"
        self.rows[(src, name, shape)] += 1
        self.total += 1
        return out


dev = torch.device("cuda")
torch.manual_seed(0)
model = build_model(zira_swint_config(device="cuda")).to(dev).train()
model.use_transformer_graph = False
if os.environ.get("FRONTEND_EAGER") == "1":     # the frozen front end (Swin, BERT) launched eagerly too: its ops become visible
    model.use_frontend_graphs = False
trainer = ZiraTrainer(model)
batches = [synthetic_batch(2, 800, 1333, n_categories=15, seed=i, device=dev) for i in range(2)]
for i in range(2):
    trainer.run_step(batches[i % 2], next_data=batches[(i + 1) % 2])
torch.cuda.synchronize()
log = Log()
with torch.autograd.set_detect_anomaly(True, check_nan=False), log:
    trainer.run_step(batches[0], next_data=batches[1])
torch.cuda.synchronize()
by_src = collections.Counter()
for (src, name, shape), n in log.rows.items():
    by_src[src] += n
print("ATen kernel-launching ops of one eager step: %d" % log.total)
print("\nby source line:")
for src, n in by_src.most_common(int(os.environ.get("TOP", 90))):
    ops = collections.Counter()
    for (s, name, shape), k in log.rows.items():
        if s == src:
            ops[name.split(".")[0]] += k
    print("x%-4d %-78s %s" % (n, src[:78], " ".join("%s:%d" % kv for kv in ops.most_common(6))))
if os.environ.get("DETAIL"):
    print("\nby (source, op, shape):")
    for (src, name, shape), n in log.rows.most_common(int(os.environ.get("TOP", 90)) * 3):
        print("x%-4d %-70s %-28s %s" % (n, src[:70], name[:28], shape))
