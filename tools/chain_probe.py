#!/usr/bin/env python3
"""GPU box: times of the window-chain kernels at the headline shape with parts of the PIPELINED kernels switched off
(csrc/gru_chain2.hpp, ChainArgs.dbg = TEMP_OPT_DEBUG >> 8): bit0 no HBM stores, bit1 no hand-over waits (both roles free-run),
bit2 no MFMAs (and no weight loads), bit3 no weight reloads (MFMAs on stale registers), bit4 matrix waves keep priority 0, bit5 gate waves at priority 3.  Results are wrong by design when a bit is set; only the kernel times matter.  Also times the two-phase kernels
of round 3 (TEMP_OPT_CHAIN_PIPELINE = 0).

    python tools/chain_probe.py [--workload S-gdelt] [--steps 5]
"""
import argparse
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="S-gdelt")
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    from temp_amd import _lib, synthetic
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    w = synthetic.workload(a.workload, seed=0)
    model = bench.build_model(w, dev)
    model.sample_rng = np.random.default_rng(2)
    targets = synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 0)
    wb = model.prepare(targets, w["L"], train=True)
    params = list(model.parameters())

    def step():
        for p in params:
            p.grad = None
        out = model.run(wb)[0]
        out.backward(torch.ones_like(out))

    def times(label):
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        tr = bench.traced_steps(step, a.steps, lib)
        f, b = tr.get("k_gru_chain_fwd", {}), tr.get("k_gru_chain_bwd", {})
        print("%-44s fwd %7.1f us   bwd %7.1f us   (timeouts %d)" % (label, 1e3 * f.get("avg_ms", 0), 1e3 * b.get("avg_ms", 0), lib.temp_gru_chain_timeouts()), flush=True)

    times("pipelined")
    for bits, what in ((1, "no HBM stores"), (4, "no MFMAs (no weight loads)"), (8, "no weight reloads"), (9, "no weight reloads, no stores"), (16, "matrix waves at priority 0"),
                       (32, "gate waves at priority 3"), (7, "no stores, no MFMAs, no waits")):
        lib.temp_set_option(_lib.OPT_DEBUG, bits << 8)
        times("pipelined, " + what)
    lib.temp_set_option(_lib.OPT_DEBUG, 0)
    lib.temp_set_option(_lib.OPT_CHAIN_PIPELINE, 0)
    times("two-phase (round 3)")
    lib.temp_set_option(_lib.OPT_CHAIN_PIPELINE, 1)


if __name__ == "__main__":
    main()
