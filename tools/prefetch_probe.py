#!/usr/bin/env python3
"""Where does a prefetched training step spend its wall time? (development probe)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from temp_amd import synthetic
from temp_amd.prefetch import BatchPrefetcher
from temp_amd.sampling import CorruptTriples
w = synthetic.workload("S-gdelt", seed=0)
dev = torch.device("cuda:0")
model = bench.build_model(w, dev)
model.sample_rng = np.random.default_rng(2)
model.corrupter = CorruptTriples(model.args, w["snapshots"], seed=5)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
batches = [synthetic.default_targets(w["num_times"], w["L"], w["bsz"], r) for r in range(45)]
prep_times = []
orig = model.prepare
def timed(*a, **k):
    t = time.perf_counter(); r = orig(*a, **k); prep_times.append(time.perf_counter() - t); return r
model.prepare = timed
mode = sys.argv[1] if len(sys.argv) > 1 else "step"
it = iter(BatchPrefetcher(model, batches, seq_len=w["L"], depth=2))
wait, work = [], []
while True:
    t0 = time.perf_counter()
    try:
        wb = next(it)
    except StopIteration:
        break
    t1 = time.perf_counter()
    if mode == "step":
        loss = model.run_loss(wb); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
    elif mode == "sync":
        loss = model.run_loss(wb); opt.zero_grad(set_to_none=True); loss.backward(); opt.step(); torch.cuda.synchronize()
    elif mode == "idle":
        time.sleep(0.005)
    t2 = time.perf_counter()
    wait.append(t1 - t0); work.append(t2 - t1)
torch.cuda.synchronize()
f = lambda x: 1e3 * float(np.mean(x[5:]))
print("mode %-5s: main waits %.2f ms for a batch, works %.2f ms; worker prepare %.2f ms per batch" % (mode, f(wait), f(work), f(prep_times)))
