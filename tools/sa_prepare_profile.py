import os, sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from temp_amd import synthetic
w = synthetic.workload("S-gdelt", seed=0)
dev = torch.device("cuda:0")
model = bench.build_model(w, dev, "attention")
model.sample_rng = np.random.default_rng(2)
batches = [synthetic.default_targets(w["num_times"], w["L"], w["bsz"], rep) for rep in range(20)]
for b in batches: model.prepare(b, w["L"], train=True)
torch.cuda.synchronize(); t0 = time.time()
for b in batches: model.prepare(b, w["L"], train=True)
torch.cuda.synchronize()
print("SA steady-state prepare: %.2f ms per batch" % (1e3 * (time.time() - t0) / len(batches)))
pr = cProfile.Profile(); pr.enable()
for b in batches: model.prepare(b, w["L"], train=True)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
