#!/usr/bin/env python3
"""The training step with the reference's default dropout (0.1 on the self-loop message, utils/args.py:17) against dropout 0:
eager launches (a step whose dropout draws is not captured), per-visit masks (no snapshot shared between windows), resident batches
(device + issue time) and a fresh batch every step through the prefetcher.  python tools/dropout_step_probe.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from temp_amd import synthetic  # noqa: E402
from temp_amd.prefetch import BatchPrefetcher  # noqa: E402
from temp_amd.sampling import CorruptTriples  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
w = synthetic.workload("S-gdelt", seed=0)
dev = torch.device("cuda:0")
for p in (0.0, 0.1):
    model = bench.build_model(w, dev)
    model.args.dropout = p
    for layer in (model.ent_encoder.layer_1, model.ent_encoder.layer_2):
        layer.dropout_p = p
    model.train()
    model.sample_rng = np.random.default_rng(2)
    model.corrupter = CorruptTriples(model.args, w["snapshots"], seed=5)
    opt = model.configure_optimizers()
    WARM = 10
    batches = [synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 1000 + r) for r in range(steps + WARM)]

    def timed(source):
        t0, edges = None, 0
        for i, wb in enumerate(source):
            if i == WARM:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            loss = model.run_loss(wb)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            if i >= WARM:
                edges += wb.n_edge_visits
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return 1e3 * dt / steps, edges / dt

    for b in batches:                                # (the first visit of a snapshot builds and uploads its cached views: once per run)
        model.prepare(b, w["L"], True)
    wbs = [model.prepare(b, w["L"], True) for b in batches[:24]]
    ms, eps = timed(wbs[i % 24] for i in range(steps + WARM))
    print("dropout %.1f  resident batches (eager, encoder + loss + backward + Adam): %.2f ms/step = %.0f M edge visits/s; distinct RGCN nodes per step %d of %d visits"
          % (p, ms, eps / 1e6, wbs[0].n_nodes_distinct, wbs[0].n_node_visits))
    ms, eps = timed(BatchPrefetcher(model, batches, seq_len=w["L"], depth=4, workers=2))
    print("dropout %.1f  fresh batch every step (prefetcher): %.2f ms/step = %.0f M edge visits/s" % (p, ms, eps / 1e6))
    from temp_amd import _lib
    lib = _lib.load()

    def one():
        loss = model.run_loss(wbs[0])
        opt.zero_grad(set_to_none=True)
        loss.backward()

    tr = bench.traced_steps(one, 3, lib)
    tot = sum(v["ms_per_step"] for v in tr.values())
    nl = sum(v["launches_per_step"] for v in tr.values())
    print("dropout %.1f  library kernels per step: %d launches, %.2f ms; top: %s" % (
        p, nl, tot, ", ".join("%s x%d %.2f ms" % (k, v["launches_per_step"], v["ms_per_step"]) for k, v in sorted(tr.items(), key=lambda kv: -kv[1]["ms_per_step"])[:8])))
    print("   batched path: %s, shared visits: %s, program: %s" % (wbs[0].batched, wbs[0].shared_visits, wbs[0].program is not None))
