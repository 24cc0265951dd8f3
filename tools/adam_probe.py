#!/usr/bin/env python3
"""Host issue time and device time of one prepared training step with torch's foreach Adam against its fused Adam (development probe)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from temp_amd import synthetic
from temp_amd.sampling import CorruptTriples
w = synthetic.workload("S-gdelt", seed=0)
dev = torch.device("cuda:0")
for fused in (False, True, False, True):
    model = bench.build_model(w, dev)
    model.sample_rng = np.random.default_rng(2)
    model.corrupter = CorruptTriples(model.args, w["snapshots"], seed=5)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=fused)
    wbs = [model.prepare(synthetic.default_targets(w["num_times"], w["L"], w["bsz"], r), w["L"], True) for r in range(22)]
    def step(wb):
        loss = model.run_loss(wb); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
    for wb in wbs[:2]: step(wb)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for wb in wbs[2:]: step(wb)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("fused=%s: issue %.3f ms/step (host), total %.3f ms/step" % (fused, 1e3 * (t1 - t0) / 20, 1e3 * (t2 - t0) / 20))
