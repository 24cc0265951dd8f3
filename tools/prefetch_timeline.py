#!/usr/bin/env python3
"""Where does a fresh-batch training step spend its wall time with the background prefetcher?  Per step: the main thread's wait for
the next prepared batch, its issue code (run_loss + backward + Adam), and -- in the worker -- the wall time of prepare() while the
main thread runs beside it (against prepare() alone: tools/prepare_stages.py).  python tools/prefetch_timeline.py [workers]"""
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

workers = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w = synthetic.workload("S-gdelt", seed=0)
dev = torch.device("cuda:0")
model = bench.build_model(w, dev)
model.sample_rng = np.random.default_rng(2)
model.corrupter = CorruptTriples(model.args, w["snapshots"], seed=5)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
batches = [synthetic.default_targets(w["num_times"], w["L"], w["bsz"], r) for r in range(65)]
for b in batches:
    model.prepare(b, w["L"], True)
prep_wall = []
orig = model.prepare


def timed_prepare(*a, **k):
    t = time.perf_counter()
    r = orig(*a, **k)
    prep_wall.append(time.perf_counter() - t)
    return r


model.prepare = timed_prepare
it = iter(BatchPrefetcher(model, batches, seq_len=w["L"], depth=2, workers=workers))
wait, issue = [], []
t_start = None
for i in range(len(batches)):
    t0 = time.perf_counter()
    wb = next(it)
    t1 = time.perf_counter()
    loss = model.run_loss(wb)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    t2 = time.perf_counter()
    if i == 5:
        torch.cuda.synchronize()
        t_start = time.perf_counter()
    if i > 5:
        wait.append(t1 - t0)
        issue.append(t2 - t1)
torch.cuda.synchronize()
n = len(wait)
print("workers %d: %.2f ms/step wall | main: wait for batch %.2f ms, issue %.2f ms | worker: prepare wall %.2f ms (median %.2f)"
      % (workers, 1e3 * (time.perf_counter() - t_start) / n, 1e3 * sum(wait) / n, 1e3 * sum(issue) / n, 1e3 * sum(prep_wall[6:]) / max(len(prep_wall) - 6, 1),
         1e3 * float(np.median(prep_wall[6:]))))
