#!/usr/bin/env python3
"""Host-side cost of issuing one training step (forward + loss + backward + Adam) for prepared batches (development probe)."""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from temp_amd import synthetic
from temp_amd.sampling import CorruptTriples
w = synthetic.workload(sys.argv[1] if len(sys.argv) > 1 else "S-gdelt", seed=0)
dev = torch.device("cuda:0")
model = bench.build_model(w, dev)
model.sample_rng = np.random.default_rng(2)
model.corrupter = CorruptTriples(model.args, w["snapshots"], seed=5)
opt = model.configure_optimizers()
wbs = [model.prepare(synthetic.default_targets(w["num_times"], w["L"], w["bsz"], r), w["L"], True) for r in range(12)]
def step(wb):
    loss = model.run_loss(wb); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
for wb in wbs[:2]: step(wb)
torch.cuda.synchronize(); t0 = time.perf_counter()
for wb in wbs[2:]: step(wb)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("issue %.2f ms/step (host), drained after another %.2f ms" % (1e3 * (t1 - t0) / 10, 1e3 * (t2 - t1)))
pr = cProfile.Profile(); pr.enable()
for wb in wbs[2:]: step(wb)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats(sys.argv[2] if len(sys.argv) > 2 else "tottime").print_stats(int(sys.argv[3]) if len(sys.argv) > 3 else 28)
