#!/usr/bin/env python3
"""Host-side cost of preparing one window batch (plan + views + upload) on the GPU box, steady state (per-snapshot caches warm):
python tools/prepare_profile.py [workload] [tottime|cumulative]"""
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

w = synthetic.workload(sys.argv[1] if len(sys.argv) > 1 else "S-gdelt", seed=0)
dev = torch.device("cuda:0")
model = bench.build_model(w, dev)
model.sample_rng = np.random.default_rng(2)
batches = [synthetic.default_targets(w["num_times"], w["L"], w["bsz"], rep) for rep in range(20)]
for b in batches:
    model.prepare(b, w["L"], train=True)
torch.cuda.synchronize()
t0 = time.time()
for b in batches:
    model.prepare(b, w["L"], train=True)
torch.cuda.synchronize()
print("steady-state prepare: %.2f ms per batch" % (1e3 * (time.time() - t0) / len(batches)))
pr = cProfile.Profile()
pr.enable()
for b in batches:
    model.prepare(b, w["L"], train=True)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats(sys.argv[2] if len(sys.argv) > 2 else "tottime").print_stats(40)
