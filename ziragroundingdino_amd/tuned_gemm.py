"""GEMM algorithm selection for the fp32 GEMMs of the step (PyTorch TunableOp).

hipBLASLt / rocBLAS pick a kernel per GEMM shape with a heuristic; for the tall fp32 GEMMs of this
model (44 446 x 256 x 2048 and friends) the heuristic's choice is 10-25 % slower than the best
kernel the libraries contain.  PyTorch's TunableOp can time the candidates once and remember the
winner per shape.  The winners for the BASELINE configs[1] step on gfx950 (ROCm 7.0 user space of
torch 2.10) are committed next to this file; ``enable()`` switches TunableOp on with tuning OFF,
so that shapes in the file use their tuned kernel and every other shape keeps the library default
-- no timing runs, no numerics beyond fp32 summation order.  The file carries validators (torch /
HIP / hipBLASLt / rocBLAS versions, GPU arch); on another software stack PyTorch ignores it.

Regenerate with ``python scripts/tune_gemms.py`` on an MI355X.
"""
import os
import tempfile

import torch

DEFAULT_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_gemm_gfx950.csv")
_state = {"enabled": False}


def enable(filename: str = DEFAULT_FILE, tune: bool = False) -> bool:
    """Returns True when TunableOp is on afterwards (needs a GPU and the results file)."""
    if not torch.cuda.is_available() or not hasattr(torch.cuda, "tunable"):
        return False
    if _state["enabled"] and not tune:
        return True
    tunable = torch.cuda.tunable
    if not tune and not os.path.exists(filename):
        return False
    tunable.enable(True)
    tunable.tuning_enable(bool(tune))
    if tune:
        tunable.set_filename(filename, insert_device_ordinal=False)
    else:
        # read the committed winners, then point the (exit-time) results file away from the package:
        # several ranks would otherwise rewrite the same file at interpreter exit
        tunable.set_filename(filename, insert_device_ordinal=False)
        try:
            tunable.read_file(filename)
        except Exception:  # validators of another software stack: keep the library defaults
            tunable.enable(False)
            return False
        scratch = os.path.join(tempfile.gettempdir(), "zira_tunableop_%d.csv" % os.getpid())
        tunable.set_filename(scratch, insert_device_ordinal=False)
    _state["enabled"] = True
    return True


def disable() -> None:
    if torch.cuda.is_available() and hasattr(torch.cuda, "tunable"):
        torch.cuda.tunable.enable(False)
    _state["enabled"] = False
