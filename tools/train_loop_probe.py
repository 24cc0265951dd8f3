#!/usr/bin/env python3
"""End-to-end training loop on the synthetic workload (development probe): every step is a NEW window batch, prepared by the
background prefetcher, then loss + backward + Adam eagerly (no graph replay).  Reports wall time per step and edge visits/s --
what a training run sees, next to bench.py's device-only number.   python tools/train_loop_probe.py [workload] [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from temp_amd import synthetic  # noqa: E402
from temp_amd.prefetch import BatchPrefetcher  # noqa: E402

w = synthetic.workload(sys.argv[1] if len(sys.argv) > 1 else "S-gdelt", seed=0)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
model = bench.build_model(w, dev, os.environ.get("PROBE_ENCODER", "gru"))
model.sample_rng = np.random.default_rng(2)
from temp_amd.sampling import CorruptTriples  # noqa: E402
model.corrupter = CorruptTriples(model.args, w["snapshots"], seed=5)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
batches = [synthetic.default_targets(w["num_times"], w["L"], w["bsz"], r) for r in range(steps + 5)]
for b in batches:                          # first visit of a snapshot builds / uploads its cached views: once per run, not per step
    model.prepare(b, w["L"], True)
for depth, label in ((2, "prefetch depth 2"), (0, "inline prepare"), (2, "prefetch depth 2")):
    it = iter(BatchPrefetcher(model, batches, seq_len=w["L"], depth=depth)) if depth else (model.prepare(b, w["L"], True) for b in batches)
    edges, t0, n = 0, None, 0
    for i, wb in enumerate(it):
        if i == 5:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        loss = model.run_loss(wb)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        if i >= 5:
            edges += wb.n_edge_visits
            n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-18s %d steps: %.2f ms/step wall, %.1f M edge visits/s (loss %.4f)" % (label, n, 1e3 * dt / n, edges / dt / 1e6, float(loss)))

if os.environ.get("PROBE_SPLIT"):                # host time of the two halves on their own (no overlap, device drained between)
    wbs, tp = [], 0.0
    for b in batches[5:25]:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        wbs.append(model.prepare(b, w["L"], True))
        tp += time.perf_counter() - t0
    tl = 0.0
    for wb in wbs:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = model.run_loss(wb)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        tl += time.perf_counter() - t0                 # launch code only: the device runs behind
    torch.cuda.synchronize()
    print("host only: prepare %.2f ms/batch, loss + backward + Adam launch code %.2f ms/step" % (1e3 * tp / len(wbs), 1e3 * tl / len(wbs)))
    for rep in range(2):                            # prepared batches, no prepare in the loop: the device time of a training step
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for wb in wbs:
            loss = model.run_loss(wb)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
        td = time.perf_counter() - t0
    print("prepared batches only (device-bound if above the launch code's time): %.2f ms/step" % (1e3 * td / len(wbs)))

if os.environ.get("PROBE_PROFILE"):
    import cProfile
    import pstats
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for b in batches[2:12]:
        wb = model.prepare(b, w["L"], True)
        loss = model.run_loss(wb)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr).sort_stats(os.environ.get("PROBE_SORT", "cumulative"))
    st.print_stats(int(os.environ.get("PROBE_TOP", "45")))
    if os.environ.get("PROBE_CALLERS"):
        st.print_callers(os.environ["PROBE_CALLERS"])

if os.environ.get("PROBE_PROFILE_MAIN"):          # the training thread's launch code alone, on prepared batches
    import cProfile
    import pstats
    wbs = [model.prepare(b, w["L"], True) for b in batches[5:25]]
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for wb in wbs:
        loss = model.run_loss(wb)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr).sort_stats(os.environ.get("PROBE_SORT", "cumulative"))
    st.print_stats(int(os.environ.get("PROBE_TOP", "45")))
