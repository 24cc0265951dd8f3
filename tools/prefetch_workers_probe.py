#!/usr/bin/env python3
"""Training loop on the synthetic workload with 1, 2 and 3 prefetch workers (development probe): wall time per step, and the
loss sequences must be identical (per-batch seeds; every kernel on the path is deterministic) -- the multi-worker run goes FIRST,
on cold snapshot caches, so first-use creation of the shared resident objects happens under several workers.
python tools/prefetch_workers_probe.py [workload] [steps]"""
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

if os.environ.get("PROBE_SWITCH_INTERVAL"):
    sys.setswitchinterval(float(os.environ["PROBE_SWITCH_INTERVAL"]))
name = sys.argv[1] if len(sys.argv) > 1 else "S-gdelt"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda:0")
w = synthetic.workload(name, seed=0)
batches = [synthetic.default_targets(w["num_times"], w["L"], w["bsz"], r) for r in range(steps + 5)]


def run(workers, warm):
    model = bench.build_model(w, dev)
    if warm:                                  # first use of every snapshot (resident views, true-set slices) outside the timed loop
        for b in batches:
            model.prepare(b, w["L"], True)
    model.sample_rng = np.random.default_rng(2)
    model.seed_rng = np.random.default_rng(3)
    model.corrupter = CorruptTriples(model.args, w["snapshots"], seed=5)
    opt = model.configure_optimizers()
    losses, t0, edges = [], None, 0
    for i, wb in enumerate(BatchPrefetcher(model, batches, seq_len=w["L"], depth=2, workers=workers, batch_seeds=True)):
        if i == 5:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        loss = model.run_loss(wb)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(loss.detach())
        if i >= 5:
            edges += wb.n_edge_visits
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return [float(x) for x in losses], 1e3 * dt / steps, edges / dt / 1e6


ref = None
order = [int(x) for x in os.environ.get("PROBE_WORKERS", "3,1,2,3,1,2,3").split(",")]
for k, workers in enumerate(order):
    losses, ms, meps = run(workers, warm=k > 0)
    same = "" if ref is None else ("  losses identical to the first run" if losses == ref else "  LOSSES DIFFER from the first run")
    ref = ref or losses
    print("workers %d%s: %.2f ms/step wall, %.1f M edge visits/s (last loss %.6f)%s"
          % (workers, " (cold caches)" if k == 0 else "", ms, meps, losses[-1], same))
