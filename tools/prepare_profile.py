#!/usr/bin/env python3
"""Host-side cost of preparing one window batch (plan + views + upload) on the GPU box: python tools/prepare_profile.py"""
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
for rep in range(12):
    targets = synthetic.default_targets(w["num_times"], w["L"], w["bsz"], rep)
    torch.cuda.synchronize()
    t0 = time.time()
    wb = model.prepare(targets, w["L"], train=True)
    torch.cuda.synchronize()
    print("prepare #%d: %.1f ms" % (rep, 1e3 * (time.time() - t0)))
pr = cProfile.Profile()
pr.enable()
wb = model.prepare(synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 3), w["L"], train=True)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
