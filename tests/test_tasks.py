"""Task chain: ``run_tasks`` -- fresh model + previous ``model_final.pth``,
train with the LR multiplier, ``after_train`` merge, save -- over two synthetic tasks, against the merged
weights the REFERENCE's modules reach through the same chain (tests/golden/gen_tasks_golden.py); plus the
checkpoint layout, the prompt-pool round trip and resuming a task from its periodic checkpoint."""
import itertools
import os

import pytest
import torch

from conftest import GOLDEN
from test_modules_golden import DEVICES, close, msda_backend  # noqa: F401
from test_train_step import build_slice_model, run_slice_step, slice_inputs

from ziragroundingdino_amd.groundingdino import GroundingDINO
from ziragroundingdino_amd.tasks import (TaskSpec, clean_state_dict, multistep_lr_multiplier, run_task,
                                         run_tasks)


class _SliceModel(GroundingDINO):
    """The slice starts after the frozen front end: a minibatch is the tuple ``slice_inputs`` makes."""

    def forward(self, data):
        return run_slice_step(self, *data)


def _specs(g, dev, tmp_path, lrs_seen):
    specs = []
    for t in g["tasks"]:
        task = t["task"]
        probe = build_slice_model(g, dev, _SliceModel, seeded=False)
        data = slice_inputs({"inputs": t["inputs"]}, probe, dev)
        specs.append(TaskSpec(name=task["name"], categories_names=task["categories"],
                              data=lambda start, d=data: itertools.repeat(d), max_iter=task["max_iter"],
                              output_dir=str(tmp_path / task["name"]),
                              lr_multiplier=multistep_lr_multiplier(task["decay_iter"])))
    return specs


@pytest.mark.parametrize("msda_backend", DEVICES, indirect=True)
def test_two_chained_tasks_match_reference(msda_backend, tmp_path):
    dev = msda_backend
    g = torch.load(os.path.join(GOLDEN, "tasks_zira_slice.pt"), weights_only=False)
    built = []

    def build_model():   # task 1 starts name-seeded; later tasks from default init + the checkpoint only
        built.append(build_slice_model(g, "cpu", _SliceModel, seeded=not built))
        return built[-1]

    totals, lrs = {}, {}

    def on_step(spec, it, loss_dict):
        totals.setdefault(spec.name, []).append(sum(loss_dict.values()))

    specs = _specs(g, dev, tmp_path, lrs)
    finals = run_tasks(specs, build_model, init_checkpoint=None, device=dev, on_step=on_step)
    assert len(built) == 2 and built[0] is not built[1]
    assert finals == [os.path.join(s.output_dir, "model_final.pth") for s in specs]

    for t, spec, final in zip(g["tasks"], specs, finals):
        for it, want in enumerate(t["totals"]):
            close(totals[spec.name][it], want, 1e-4, "%s total loss at iteration %d" % (spec.name, it))
        ck = torch.load(final, weights_only=False)
        assert set(ck) == {"model", "trainer", "iteration"} and ck["iteration"] == spec.max_iter
        for n, want in t["merged"].items():
            close(ck["model"][n], want, 1e-4, "%s merged %s" % (spec.name, n))
        with open(os.path.join(spec.output_dir, "last_checkpoint")) as f:
            assert f.read() == "model_final.pth"
        # the periodic checkpoint of the last iteration holds the weights BEFORE the merge
        pre = torch.load(os.path.join(spec.output_dir, "model_%07d.pth" % (spec.max_iter - 1)), weights_only=False)
        for n, want in t["before_rep"].items():
            close(pre["model"][n], want, 1e-4, "%s before merge %s" % (spec.name, n))
        assert pre["trainer"]["iteration"] == spec.max_iter - 1 and pre["trainer"]["optimizer"]["state"]

    # prompt pool: entries of both tasks under the reference's key names, and they survive the load
    last = torch.load(finals[-1], weights_only=False)["model"]
    names = [c for t in g["tasks"] for c in t["task"]["categories"]]
    assert [k for k in last if k.startswith("prompt_memory_pool.")] == ["prompt_memory_pool.-%s-" % c for c in names]
    first = torch.load(finals[0], weights_only=False)["model"]
    for c in g["tasks"][0]["task"]["categories"]:     # task 2 neither trains nor re-draws task 1's entries
        assert torch.equal(first["prompt_memory_pool.-%s-" % c], last["prompt_memory_pool.-%s-" % c])
    model = build_slice_model(g, "cpu", _SliceModel, seeded=False)
    model.load_state_dict(clean_state_dict({"module." + k: v for k, v in last.items()}), strict=True)
    assert model.learned_classes == names
    # after the merge every branch is back at its start: 1e-8 weights, scaling at its init
    assert torch.all(model.rep_linear_adapter.weight == 1e-8) and float(model.rep_linear_adapter.scaling.detach()) == pytest.approx(0.1)
    assert model.rep_linear_adapter.freeze_linear.weight.abs().max() > 1e-4


def test_resume_continues_from_periodic_checkpoint(tmp_path, oracle):
    """A task cut after its periodic checkpoint and resumed ends with the same weights as an
    uninterrupted one (optimizer moments and iteration travel in the checkpoint)."""
    g = torch.load(os.path.join(GOLDEN, "tasks_zira_slice.pt"), weights_only=False)
    t = g["tasks"][0]
    data = slice_inputs({"inputs": t["inputs"]}, None, "cpu")
    build = lambda: build_slice_model(g, "cpu", _SliceModel)

    def spec(out, max_iter):
        return TaskSpec(name="a", categories_names=["fish"], data=lambda start: itertools.repeat(data),
                        max_iter=max_iter, output_dir=str(tmp_path / out), checkpoint_period=1,
                        lr_multiplier=multistep_lr_multiplier(2))

    whole = torch.load(run_task(spec("whole", 4), build, None), weights_only=False)["model"]

    class _Cut(Exception):
        pass

    def cut(spec_, it, loss_dict):
        if it == 3:
            raise _Cut()

    with pytest.raises(_Cut):   # cut AFTER the decay iteration: the checkpoint holds the multiplied learning rates
        run_task(spec("cut", 4), build, None, on_step=cut)
    assert sorted(os.listdir(tmp_path / "cut")) == ["last_checkpoint", "model_0000000.pth", "model_0000001.pth",
                                                    "model_0000002.pth"]
    lrs = [g["lr"] for g in torch.load(tmp_path / "cut" / "model_0000002.pth", weights_only=False)["trainer"]["optimizer"]["param_groups"]]
    assert max(lrs) == pytest.approx(1e-4)
    seen = []
    final = run_task(spec("cut", 4), build, None, resume=True, on_step=lambda s, it, l: seen.append(it))
    assert seen == [3]
    resumed = torch.load(final, weights_only=False)["model"]
    for n, v in whole.items():
        if "adapter" in n:
            close(resumed[n], v, 1e-6, "resumed " + n)
    # a finished task is not trained again
    assert run_task(spec("cut", 4), build, None, resume=True, on_step=lambda *a: seen.append("again")) == final
    assert seen == [3]


def _chain_worker(rank, world, port, gpath, out_dir):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    g = torch.load(gpath, weights_only=False)
    built = []

    def build_model():
        built.append(build_slice_model(g, "cpu", _SliceModel, seeded=not built))
        return built[-1]

    specs = []
    for t in g["tasks"]:
        inp, feats, poss, am, pid, c2t = slice_inputs({"inputs": t["inputs"]}, None, "cpu")
        if world > 1:       # each rank trains on one image of the two-image minibatch
            sl = slice(rank, rank + 1)
            inp = dict(inp, bert_hidden=inp["bert_hidden"][sl], input_ids=inp["input_ids"][sl],
                       img_mask=inp["img_mask"][sl], targets=inp["targets"][sl])
            from ziragroundingdino_amd.utils import NestedTensor
            feats = [NestedTensor(f.tensors[sl], f.mask[sl]) for f in feats]
            poss, am, pid, c2t = [p[sl] for p in poss], am[sl], pid[sl], c2t[sl]
        data = (inp, feats, poss, am, pid, c2t)
        task = t["task"]
        specs.append(TaskSpec(name=task["name"], categories_names=task["categories"],
                              data=lambda start, d=data: itertools.repeat(d), max_iter=task["max_iter"],
                              output_dir=os.path.join(out_dir, task["name"]),
                              lr_multiplier=multistep_lr_multiplier(task["decay_iter"])))
    finals = run_tasks(specs, build_model, device="cpu")
    assert all(os.path.exists(f) for f in finals)          # rank 0 wrote them before the barrier released us
    dist.destroy_process_group()


def test_task_chain_two_ranks_gloo(tmp_path):
    """The chain under data parallelism (world_size 2, gloo, one image per rank): rank 0 writes the checkpoints,
    every rank starts the next task from them, and the merged weights equal the one-process run on both images
    (the averaged side-branch gradients are those of the whole minibatch)."""
    import torch.multiprocessing as mp

    gpath = os.path.join(GOLDEN, "tasks_zira_slice.pt")
    port = 29800 + os.getpid() % 1500
    mp.spawn(_chain_worker, args=(2, port, gpath, str(tmp_path / "w2")), nprocs=2, join=True)
    g = torch.load(gpath, weights_only=False)
    for t in g["tasks"]:
        ck = torch.load(tmp_path / "w2" / t["task"]["name"] / "model_final.pth", weights_only=False)["model"]
        for n, want in t["merged"].items():
            close(ck[n], want, 2e-4, "2-rank %s merged %s" % (t["task"]["name"], n))


def test_bench_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` outside torchrun hands over to torch.distributed.run on 127.0.0.1 before anything
    touches the GPU (reference train_multidatasets.py:573-580 launches its ranks itself too)."""
    import subprocess
    import sys

    import bench

    calls = {}
    monkeypatch.setattr(bench, "visible_gpus", lambda: 4)   # (counted from the environment / sysfs, never through HIP)
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: calls.update(cmd=cmd, env=env) or 0)
    with pytest.raises(SystemExit) as e:
        bench.spawn_ranks(4, ["--gpus", "4", "--steps", "3"])
    assert e.value.code == 0
    cmd = calls["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    with pytest.raises(SystemExit, match="only 4 GPU"):
        bench.spawn_ranks(8, [])
