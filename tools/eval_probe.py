#!/usr/bin/env python3
"""Wall time of evaluate() (window encoder on full graphs + filtered ranking of every triple of the target snapshots) on the
synthetic workloads (development probe):  python tools/eval_probe.py [workload]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from temp_amd import synthetic
w = synthetic.workload(sys.argv[1] if len(sys.argv) > 1 else "S-gdelt", seed=0)
dev = torch.device("cuda:0")
model = bench.build_model(w, dev, os.environ.get("PROBE_ENCODER", "gru"))
batches = [torch.tensor(synthetic.default_targets(w["num_times"], w["L"], w["bsz"], r)) for r in range(12)]
for i, b in enumerate(batches):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ranks, loss = model.evaluate(b, val=True)
    torch.cuda.synchronize()
    if i >= 2:
        print("evaluate: %.1f ms for %d ranks (MRR %.4f)" % (1e3 * (time.perf_counter() - t0), ranks.numel(), float((1.0 / ranks.float()).mean())))
