"""CPU-side checks of the drop-in boundary: the shared library loads and exports exactly the
symbols include/zira_msda.h declares, the `_C` drop-in reproduces the reference's error
behaviour for CPU tensors, and host-side planning helpers answer without a GPU."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT

from ziragroundingdino_amd import _C, _lib
from ziragroundingdino_amd import build as zbuild


def header_symbols():
    text = open(os.path.join(ROOT, "include", "zira_msda.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zira_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    zbuild.build_extension()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = header_symbols()
    assert declared, "no declarations parsed from include/zira_msda.h"
    for sym in declared:
        assert hasattr(lib, sym), "libzira_msda.so does not export %s" % sym
    assert sorted(_lib.SYMBOLS) == declared      # the binding knows exactly the header's surface


def test_host_only_entry_points_work_without_gpu():
    lib = _lib.load()
    assert _lib.version().startswith("zira_msda")
    assert _lib.variant_f32(32) and _lib.variant_f32(24) == "generic"
    # north-star decoder shape: the atomic-free backward applies and needs a few MB of scratch
    n = lib.zira_msda_bwd_workspace_bytes(2, 22223, 8, 32, 4, 900, 4)
    assert 1 << 20 < n < 1 << 28
    assert lib.zira_msda_bwd_workspace_bytes(2, 22223, 8, 24, 4, 900, 4) == 0   # D=24: generic path only
    assert lib.zira_msda_bwd_workspace_bytes(0, 1, 1, 32, 1, 1, 1) == 0
    assert lib.zira_rsb_workspace_floats(1 << 20) > 0
    # argument errors are reported, never thrown, and nothing is launched
    assert lib.zira_msda_fwd_f32(None, None, None, None, None, 1, 1, 1, 32, 1, 1, 1, None, None) == 1


def test_cpu_tensors_raise_like_the_reference():
    v = torch.zeros(1, 4, 2, 32)
    sh = torch.tensor([[2, 2]])
    st = torch.tensor([0])
    loc = torch.zeros(1, 3, 2, 1, 4, 2)
    attn = torch.zeros(1, 3, 2, 1, 4)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        _C.ms_deform_attn_forward(v, sh, st, loc, attn, 64)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        _C.ms_deform_attn_backward(v, sh, st, loc, attn, torch.zeros(1, 3, 64), 64)


def test_missing_extension_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()
