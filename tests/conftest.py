import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_msda_cases():
    return sorted(glob.glob(os.path.join(GOLDEN, "msda_*.npz")))


def load_npz(path):
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def oracle():
    from oracle import msda_oracle

    msda_oracle.build()
    return msda_oracle


@pytest.fixture(autouse=True)
def _quiesce_gpu_between_tests(request):
    """GPU tests build models with hipGraphs, side streams and multi-GB caches.  What a test leaves behind is collected
    and the device drained BEFORE the next test starts, so that graph / stream / event teardown never happens at a random
    allocation of a later test while that test has work in flight on several streams."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    import gc

    import torch

    if torch.cuda.is_available():
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.synchronize()
