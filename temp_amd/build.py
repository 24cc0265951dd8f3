"""Build libtemp_amd.so (hand-written HIP kernels + C ABI) for gfx950 and libtemp_host.so (host planner, g++), in-tree.

    python -m temp_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with gpurun snapshots.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libtemp_amd.so")
SOURCES = ["rgcn_kernels.hip", "gemm_kernels.hip", "gru_kernels.hip", "attn_kernels.hip"]
HEADERS = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".hpp")] + [os.path.join(REPO, "include", "temp_amd.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    from . import _hostlib
    _hostlib.build(force=force, verbose=verbose)            # host planner (plain C++, g++): temp_amd/libtemp_host.so
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-comment",
           "-I" + os.path.join(REPO, "include"), "-I" + CSRC, "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print("[temp_amd.build] " + " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
