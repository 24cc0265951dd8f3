#!/usr/bin/env python3
"""Host side of ONE training step over already prepared batches (run_loss + backward + Adam, eager launches): wall per step with
the device drained each step (= host issue time + device tail), host-only issue time (no sync), and a cProfile of the issue code.
python tools/step_host_profile.py [workload] [tottime|cumulative]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from temp_amd import synthetic  # noqa: E402
from temp_amd.sampling import CorruptTriples  # noqa: E402

w = synthetic.workload(sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("t") and not sys.argv[1].startswith("c") else "S-gdelt", seed=0)
key = [a for a in sys.argv[1:] if a in ("tottime", "cumulative")]
dev = torch.device("cuda:0")
model = bench.build_model(w, dev)
model.sample_rng = np.random.default_rng(2)
model.corrupter = CorruptTriples(model.args, w["snapshots"], seed=5)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
batches = [synthetic.default_targets(w["num_times"], w["L"], w["bsz"], r) for r in range(24)]
wbs = [model.prepare(b, w["L"], True) for b in batches]


def step(wb):
    loss = model.run_loss(wb)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for wb in wbs[:4]:
    step(wb)
torch.cuda.synchronize()
t0 = time.perf_counter()
for wb in wbs[4:]:
    step(wb)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
n = len(wbs) - 4
print("prepared batches, eager step: host issue %.2f ms/step, wall %.2f ms/step (device-bound when wall > issue)" % (1e3 * t_issue / n, 1e3 * t_all / n))
pr = cProfile.Profile()
pr.enable()
for wb in wbs[4:]:
    step(wb)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats(key[0] if key else "tottime").print_stats(45)
