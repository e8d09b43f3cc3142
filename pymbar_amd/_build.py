"""Build libmbar_hip.so (gfx950 only) in-tree with hipcc.  Used by ``__graft_entry__.build()``.

The shared object is git-ignored but travels with the working tree; nothing is compiled at
import time and there is no fallback if it is missing (see ``pymbar_amd/_lib.py``).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libmbar_hip.so")
# Kernel families in separate translation units (compiled in parallel: ~40 s on 8 cores instead of 2.5 min for the one file of
# rounds 1-3; an edit to one family recompiles that family): shared device helpers in mbar_device.h.
KERNEL_SOURCES = ["mbar_k_eval.hip", "mbar_k_gram.hip", "mbar_k_quad.hip", "mbar_k_pmode.hip", "mbar_k_fused.hip", "mbar_k_solver.hip"]
HOST_SOURCES = ["mbar_capi.cpp", "mbar_loops.cpp", "mbar_comm.cpp", "mbar_host.cpp"]  # (shared internal header: mbar_ctx.h)
SOURCES = KERNEL_SOURCES + HOST_SOURCES
COMMON_DEPS = ["mbar_internal.h", os.path.join("..", "..", "include", "mbar_hip.h")]
HOST_DEPS = ["mbar_ctx.h"]
KERNEL_DEPS = ["mbar_device.h", "exp2_table.inc", "log_table.inc"]
DEPS = SOURCES + COMMON_DEPS + KERNEL_DEPS + HOST_DEPS
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# (A/B builds of kernel variants kept behind macros: MBAR_EXTRA_HIPCC_FLAGS="-DMBAR_..." python -m pymbar_amd._build --force)
FLAGS += os.environ.get("MBAR_EXTRA_HIPCC_FLAGS", "").split()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in deps)


def _compile(src):
    obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
    deps = [src] + COMMON_DEPS + (KERNEL_DEPS if src in KERNEL_SOURCES else HOST_DEPS)
    if _stale(obj, deps):
        cmd = [HIPCC] + FLAGS + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        subprocess.run(cmd, check=True, cwd=CSRC)
    return obj


def build_library(force=False, verbose=True):
    """Compile the HIP sources for gfx950 and link ``csrc/libmbar_hip.so``."""
    if force:
        for f in os.listdir(CSRC):
            if f.endswith((".o", ".so")):
                os.remove(os.path.join(CSRC, f))
    if not _stale(LIB, DEPS) and not force:
        return LIB
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 2)) as ex:
        objs = list(ex.map(_compile, SOURCES))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    subprocess.run(cmd, check=True, cwd=CSRC)
    if verbose:
        print(f"built {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
