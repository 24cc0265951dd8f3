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
# Kernels whose inline-assembly loads stay in flight across many instructions (gru_chain2.hpp): a register the allocator spills
# there would be stored before its data has arrived.  Their resource report is kept next to the object and checked: 0 spills.
RESOURCE_REPORT = {"gru_chain.hip": ("k_gru_chain_fwd2", "k_gru_chain_bwd2")}


def _check_resources(src_name, report_path):
    """Parse a -Rpass-analysis=kernel-resource-usage report; raise when a guarded kernel spills vector registers."""
    import re
    bad, name = [], None
    for line in open(report_path, errors="replace"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"VGPRs Spill: (\d+)", line)
        if m and name and any(k in name for k in RESOURCE_REPORT[src_name]) and int(m.group(1)) > 0:
            bad.append((name, int(m.group(1))))
    if bad:
        raise RuntimeError("vector-register spills in kernels with asynchronous inline-assembly loads: %s" % bad)


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
            report = None
            if s in RESOURCE_REPORT:
                cmd.append("-Rpass-analysis=kernel-resource-usage")
                report = open(obj + ".resources.txt", "w")
            if verbose:
                print("[temp_amd.build] " + " ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd, stderr=report), s, report))
    for cmd, p, s, report in procs:
        rc = p.wait()
        if report is not None:
            report.close()
        if rc != 0:
            if report is not None:
                sys.stderr.write(open(report.name, errors="replace").read()[-4000:])
            raise subprocess.CalledProcessError(rc, cmd)
        if report is not None:
            _check_resources(s, report.name)
    if procs or force or _newer(objs, LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[temp_amd.build] " + " ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
