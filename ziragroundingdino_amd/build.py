"""In-tree build of the HIP extension (``libzira_msda.so``) for gfx950.

``hipcc`` cross-compiles without a GPU; the shared object is written next to this file so it
travels with the tree (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "libzira_msda.so")
SOURCES = [os.path.join(_HERE, "csrc", f) for f in ("msda.hip", "msda_cells.hip", "msda_tiles.hip", "msda_cpu.cpp", "rsb.hip", "xty.hip", "bisoftmax.hip", "layernorm.hip", "lsap.hip", "catlogits.hip", "winattn.hip", "refpoints.hip", "attn.hip", "sampling.hip", "gemm_drelu.hip", "rowgemm.hip", "gemm_bf16x3.hip", "criterion.hip", "textside.hip")]
HEADERS = [os.path.join(_ROOT, "include", "zira_msda.h"), os.path.join(_HERE, "csrc", "msda_internal.h")]
HIPCC_FLAGS = [
    "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
    "-munsafe-fp-atomics",       # fp32 atomics -> global_atomic_add_f32 (no CAS loop)
    "-I" + os.path.join(_ROOT, "include"), "-pthread",
]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the HIP extension cannot be built")
    return exe


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS + [__file__])


def build_extension(force=False, verbose=False):
    """Compile every HIP source into libzira_msda.so (no-op when up to date)."""
    if not force and not needs_build():
        return LIB_PATH
    cmd = [_hipcc()] + HIPCC_FLAGS + SOURCES + ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_extension(force=True, verbose=True))
