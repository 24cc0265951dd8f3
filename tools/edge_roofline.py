#!/usr/bin/env python3
"""HBM-roofline check of the edge kernels on ONE snapshot of the S-hbm shape (2^20 nodes, 2^24 edges,
230 relations, D = 200 -- working set >> the 256 MB Infinity Cache), where S-gdelt's 400 KB node matrix is
cache-resident and says nothing about HBM.  Times temp_rgcn_fwd / temp_rgcn_bwd per kernel with the library's
event trace and prints algorithmic GB/s against the 8 TB/s peak.

    python tools/edge_roofline.py [--log2-nodes 20] [--log2-edges 24] [--reps 5] > profiles/r01_edge_kernels_shbm.json
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from temp_amd import _lib, synthetic  # noqa: E402
from temp_amd import backend as TB  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-nodes", type=int, default=20)
    ap.add_argument("--log2-edges", type=int, default=24)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--relations", type=int, default=230, help="20: the relation weights fit LDS (the S-gdelt kernels on the HBM-sized graph)")
    a = ap.parse_args()
    n, E, R, D, B = 1 << a.log2_nodes, 1 << a.log2_edges, a.relations, 200, 100
    dev = torch.device("cuda:0")
    t0 = time.time()
    g = synthetic.make_snapshots(n, R, E, n, 1, seed=0)[0]
    dg = g.device_graph(dev, 2 * R)
    prep = time.time() - t0
    be, lib = TB.get_backend(), _lib.load()
    gen = torch.Generator(device="cpu").manual_seed(2)
    S = D // B
    w = (torch.rand(2 * R, B * S * S, generator=gen) - 0.5).to(dev)
    lw = ((torch.rand(D, D, generator=gen) - 0.5) * 0.2).to(dev)
    h = torch.randn(n, D, device=dev)
    gy = torch.randn(n, D, device=dev)
    out = be.rgcn_fwd(dg, h, None, w, lw, None, B, 1)
    be.rgcn_bwd(dg, h, out, gy, w, lw, False, B, 1)
    torch.cuda.synchronize()
    cap = 4096
    _lib.check(lib.temp_trace_begin(cap), "trace_begin")
    for _ in range(a.reps):
        out = be.rgcn_fwd(dg, h, None, w, lw, None, B, 1)
        be.rgcn_bwd(dg, h, out, gy, w, lw, False, B, 1)
    ids, ms, cnt = (ctypes.c_int32 * cap)(), (ctypes.c_float * cap)(), ctypes.c_int32(0)
    _lib.check(lib.temp_trace_end(ids, ms, cap, ctypes.byref(cnt)), "trace_end")
    agg = {}
    for i in range(cnt.value):
        k = lib.temp_trace_kernel_name(ids[i]).decode()
        v = agg.setdefault(k, [0, 0.0])
        v[0] += 1
        v[1] += ms[i]
    row = 4 * D
    alg = {"k_rgcn_agg<fwd>": E * (row + 8) + n * row, "k_rgcn_agg<dx>": E * (row + 12) + n * row, "k_rgcn_dw": E * (2 * row + 12),
           "k_gemm_panel<loop_fwd>": 3 * n * row, "k_gemm_panel<loop_dx>": 3 * n * row, "k_relu_bwd": 3 * n * row}
    res = {}
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        avg = t / c
        e = dict(launches=c // a.reps, avg_ms=avg)
        if k in alg:
            e["algorithmic_bytes"] = alg[k]
            e["GBps"] = alg[k] / (avg * 1e-3) / 1e9
            e["frac_of_8TBps"] = e["GBps"] / 8000.0
        res[k] = e
    print(json.dumps(dict(shape=dict(nodes=n, edges=E, relations=R, D=D, n_bases=B), host_prepare_s=prep, kernels=res), indent=1))


if __name__ == "__main__":
    main()
