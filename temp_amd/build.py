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
SOURCES = ["rgcn_kernels.hip", "gemm_kernels.hip", "gru_kernels.hip", "gru_chain.hip", "attn_kernels.hip", "store_kernels.hip"]
HEADERS = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".hpp")] + [os.path.join(REPO, "include", "temp_amd.h")]
OBJDIR = os.path.join(CSRC, "build")
# per-source flags (none at present; -fno-slp-vectorize on rgcn_kernels.hip -- v_fmac_f32 pairs instead of v_pk_fma_f32 in the
# edge kernels -- measured neutral with tools/tile_phases.py)
EXTRA_FLAGS = {}

def _newer(deps, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """One object per .hip source (recompiled only when it or a header changed, in parallel), then one link."""
    from . import _hostlib
    _hostlib.build(force=force, verbose=verbose)            # host planner (plain C++, g++): temp_amd/libtemp_host.so
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJDIR, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment", "-I" + os.path.join(REPO, "include"), "-I" + CSRC]
    objs, procs = [], []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer([src] + HEADERS, obj):
            cmd = [hipcc] + flags + EXTRA_FLAGS.get(s, []) + ["-c", src, "-o", obj]
            if verbose:
                print("[temp_amd.build] " + " ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    if procs or force or _newer(objs, LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[temp_amd.build] " + " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
