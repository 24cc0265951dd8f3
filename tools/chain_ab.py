#!/usr/bin/env python3
"""GPU box: A/B of the window-chain kernels' development switches on ONE box (boxes differ by a few per cent), headline shape.
ChainArgs.dbg = TEMP_OPT_DEBUG >> 8: bit 6 every block starts its W_hh slab walk at slab 0 (round 3), bit 7 a wave's tile slot past
the last tile loads a duplicate tile's planes (round 3).  Then the input gates once per distinct row (GruProgram.gi_shared) against
the gates of every chain row.  (Non-temporal loads / stores of the row streams were tried the same way: +-3 %, inside the run-to-run
spread -- not kept.)   python tools/chain_ab.py [--steps 5]"""
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
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    from temp_amd import _lib, synthetic
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    w = synthetic.workload("S-gdelt", seed=0)
    model = bench.build_model(w, dev)
    model.sample_rng = np.random.default_rng(2)
    wb = model.prepare(synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 0), w["L"], train=True)
    st = bench.GraphStep(lambda: model.run(wb)[0], list(model.parameters()), graph=False)
    for bits, what in ((0, "product"), (64, "no slab rotation"), (128, "duplicate-tile loads"), (192, "round 3 (both)"), (0, "product (again)")):
        lib.temp_set_option(_lib.OPT_DEBUG, bits << 8)
        for _ in range(2):
            st.eager()
        torch.cuda.synchronize()
        tr = bench.traced_steps(st.eager, a.steps, lib)
        print("%-26s fwd %7.1f us   bwd %7.1f us" % (what, 1e3 * tr["k_gru_chain_fwd"]["avg_ms"], 1e3 * tr["k_gru_chain_bwd"]["avg_ms"]), flush=True)
    lib.temp_set_option(_lib.OPT_DEBUG, 0)
    # input gates once per distinct row (GruProgram.gi_shared) against gates of every chain row
    prog = wb.program
    src = prog.x_src
    for share in (True, False, True):
        prog.__dict__.pop("_gi_shared", None)
        if share:
            prog.x_src = src
        else:
            prog.__dict__.pop("x_src", None)
        for _ in range(2):
            st.eager()
        torch.cuda.synchronize()
        tr = bench.traced_steps(st.eager, a.steps, lib)
        gi = [v for k, v in tr.items() if "gru_gi" in k]
        print("shared input gates %-5s  fwd %7.1f us   gi GEMM %7.1f us   step kernels %.3f ms" % (
            share, 1e3 * tr["k_gru_chain_fwd"]["avg_ms"], 1e3 * sum(v["ms_per_step"] for v in gi), sum(v["ms_per_step"] for v in tr.values())), flush=True)


def weight_grads_ab(steps=200):
    """Both directions' weight gradients in one launch (temp_gru_weight_grads_multi) against one set of launches per GRU: the whole
    step as a HIP graph, alternating, on this box."""
    from temp_amd import gru_chain as GC, synthetic
    dev = torch.device("cuda", 0)
    w = synthetic.workload("S-gdelt", seed=0)
    model = bench.build_model(w, dev)
    model.sample_rng = np.random.default_rng(2)
    wb = model.prepare(synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 0), w["L"], train=True)
    res = {}
    for rep in range(2):
        for flag in (True, False):
            GC.WEIGHT_GRADS_MULTI = flag
            st = bench.GraphStep(lambda: model.run(wb)[0], list(model.parameters()), graph=True)
            res.setdefault(flag, []).append(st.time(steps, 20))
    GC.WEIGHT_GRADS_MULTI = True
    for flag, v in res.items():
        print("weight gradients of both directions in one launch: %-5s  step %s ms (graph replays)" % (flag, ", ".join("%.3f" % x for x in v)), flush=True)


def gate_grads_ab(steps=200):
    """Gate gradients written once (temp_gru_chain_bwd_g4 + temp_gru_grads_g4: k_gru_wgrad) against dgi + dgh (temp_gru_chain_bwd +
    temp_gru_weight_grads_multi): the whole step as a HIP graph, alternating, on this box; then the kernels of an eager step."""
    from temp_amd import _lib, gru_chain as GC, synthetic
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    w = synthetic.workload("S-gdelt", seed=0)
    model = bench.build_model(w, dev)
    model.sample_rng = np.random.default_rng(2)
    wb = model.prepare(synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 0), w["L"], train=True)
    res = {}
    for rep in range(3):
        for flag in (True, False):
            GC.GATE_GRADS_ONCE = flag
            st = bench.GraphStep(lambda: model.run(wb)[0], list(model.parameters()), graph=True)
            res.setdefault(flag, []).append(st.time(steps, 20))
    for flag, v in res.items():
        print("gate gradients once: %-5s  step %s ms (graph replays)" % (flag, ", ".join("%.3f" % x for x in v)), flush=True)
    for flag in (True, False):
        GC.GATE_GRADS_ONCE = flag
        st = bench.GraphStep(lambda: model.run(wb)[0], list(model.parameters()), graph=False)
        for _ in range(2):
            st.eager()
        torch.cuda.synchronize()
        tr = bench.traced_steps(st.eager, 5, lib)
        keys = ("k_gru_chain_bwd", "k_gru_wgrad", "k_gemm_tn_bx8", "k_gemm_panel<gru_dx>", "k_reduce_slices")
        print("gate gradients once: %-5s  " % flag + "  ".join("%s %.1f us" % (k, 1e3 * tr[k]["ms_per_step"]) for k in keys if k in tr)
              + "  all kernels %.3f ms" % sum(v["ms_per_step"] for v in tr.values()), flush=True)
    GC.GATE_GRADS_ONCE = True


def debug_ab(value, what, steps=200):
    """The whole step as a HIP graph with TEMP_OPT_DEBUG = 0 against `value` (a development switch of the library), alternating."""
    from temp_amd import _lib, synthetic
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    w = synthetic.workload("S-gdelt", seed=0)
    model = bench.build_model(w, dev)
    model.sample_rng = np.random.default_rng(2)
    wb = model.prepare(synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 0), w["L"], train=True)
    res = {}
    for rep in range(2):
        for v in (0, value):
            lib.temp_set_option(_lib.OPT_DEBUG, v)
            st = bench.GraphStep(lambda: model.run(wb)[0], list(model.parameters()), graph=True)
            res.setdefault(v, []).append(st.time(steps, 20))
    lib.temp_set_option(_lib.OPT_DEBUG, 0)
    for v, t in res.items():
        print("%s: TEMP_OPT_DEBUG %-4d step %s ms (graph replays)" % (what, v, ", ".join("%.3f" % x for x in t)), flush=True)


if __name__ == "__main__":
    if "--debug" in sys.argv:
        i = sys.argv.index("--debug")
        debug_ab(int(sys.argv[i + 1]), " ".join(sys.argv[i + 2:]) or "switch")
        sys.exit(0)
    if "--gate-grads" in sys.argv:
        gate_grads_ab()
        sys.exit(0)
    if "--weight-grads" in sys.argv:
        sys.argv.remove("--weight-grads")
        weight_grads_ab()
        sys.exit(0)
    main()
