"""In-tree build of the HIP extension (``libzira_msda.so``) for gfx950.

``hipcc`` cross-compiles without a GPU; the shared object is written next to this file so it
travels with the tree (it is git-ignored, not gpurun-ignored).  Every source is compiled to its own
object under ``csrc/_obj`` (only what changed, a few at a time) and the objects are linked.
"""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "libzira_msda.so")
OBJ_DIR = os.path.join(_HERE, "csrc", "_obj")
SOURCES = [os.path.join(_HERE, "csrc", f) for f in ("msda.hip", "msda_cells.hip", "msda_tiles.hip", "msda_cpu.cpp", "rsb.hip", "xty.hip", "xty_bf16x3.hip", "bisoftmax.hip", "layernorm.hip", "groupnorm.hip", "lsap.hip", "catlogits.hip", "winattn.hip", "refpoints.hip", "attn.hip", "sampling.hip", "gemm_drelu.hip", "rowgemm.hip", "gemm_bf16x3.hip", "gemm_f16x2.hip", "gemm_f16x2_panel.hip", "ffn_f16x2.hip", "thin_f16x2.hip", "criterion.hip", "textside.hip")]
HEADERS = [os.path.join(_ROOT, "include", "zira_msda.h"), os.path.join(_HERE, "csrc", "msda_internal.h"),
           os.path.join(_HERE, "csrc", "msda_fwd_lean.h")]
HIPCC_FLAGS = [
    "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC",
    "-munsafe-fp-atomics",       # fp32 atomics -> global_atomic_add_f32 (no CAS loop)
    "-I" + os.path.join(_ROOT, "include"), "-pthread",
]
# per-source additions
EXTRA_FLAGS = {
    # packed fp32 vector instructions beside matrix instructions cost more than the two plain ones they replace
    "ffn_f16x2.hip": ["-fno-slp-vectorize"],
}
JOBS = 4


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; the HIP extension cannot be built")
    return exe


def _obj(src):
    return os.path.join(OBJ_DIR, os.path.basename(src) + ".o")


def _stale(src, newest_common):
    o = _obj(src)
    return not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(src), newest_common)


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS + [__file__])


def build_extension(force=False, verbose=False):
    """Compile every HIP source into libzira_msda.so (no-op when up to date)."""
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    common = max(os.path.getmtime(p) for p in HEADERS + [__file__])
    hipcc = _hipcc()

    def compile_one(src):
        cmd = [hipcc] + HIPCC_FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", _obj(src) + ".tmp"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(_obj(src) + ".tmp", _obj(src))

    todo = [s for s in SOURCES if force or _stale(s, common)]
    with ThreadPoolExecutor(max_workers=JOBS) as pool:
        list(pool.map(compile_one, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + [_obj(s) for s in SOURCES] + ["-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_extension(force=True, verbose=True))
