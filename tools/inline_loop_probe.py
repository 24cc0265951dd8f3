#!/usr/bin/env python3
"""Inline training loop (prepare, then the step, on one thread): host time of `prepare` and of the step's issue INSIDE the loop,
i.e. with the previous step's kernels still running (development probe)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from temp_amd import synthetic
from temp_amd.sampling import CorruptTriples
w = synthetic.workload("S-gdelt", seed=0)
dev = torch.device("cuda:0")
model = bench.build_model(w, dev)
model.sample_rng = np.random.default_rng(2)
model.corrupter = CorruptTriples(model.args, w["snapshots"], seed=5)
opt = model.configure_optimizers()
N = 120
batches = [synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 1000 + r) for r in range(N)]
for b in batches: model.prepare(b, w["L"], True)
tp = ti = 0.0
torch.cuda.synchronize()
for i, b in enumerate(batches):
    if i == 20:
        torch.cuda.synchronize(); t0 = time.perf_counter(); tp = ti = 0.0
    a = time.perf_counter()
    wb = model.prepare(b, w["L"], True)
    c = time.perf_counter()
    loss = model.run_loss(wb); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
    d = time.perf_counter()
    tp += c - a; ti += d - c
torch.cuda.synchronize()
tot = time.perf_counter() - t0
n = N - 20
print("inline loop: %.2f ms/step wall; prepare %.2f ms, issue %.2f ms (host, inside the loop)" % (1e3 * tot / n, 1e3 * tp / n, 1e3 * ti / n))
