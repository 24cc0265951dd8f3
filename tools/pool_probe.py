#!/usr/bin/env python3
"""Does a pool of prepare() workers scale past the GIL? (development probe)"""
import os, sys, time, threading
from concurrent.futures import ThreadPoolExecutor
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
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
batches = [synthetic.default_targets(w["num_times"], w["L"], w["bsz"], r) for r in range(45)]
for b in batches:                      # warm the per-snapshot caches (first-visit costs are per epoch, not per step)
    model.prepare(b, w["L"], True)
sys.setswitchinterval(2e-4)
tl = threading.local()
def job(b):
    st = getattr(tl, "st", None)
    if st is None:
        st = tl.st = torch.cuda.Stream(dev)
    with torch.cuda.stream(st):
        wb = model.prepare(b, w["L"], True)
        ev = torch.cuda.Event(); ev.record(st)
    return wb, ev
for workers in (1, 2, 3, 4):
    for do_step in (False, True):
        with ThreadPoolExecutor(workers) as ex:
            torch.cuda.synchronize(); t0 = time.perf_counter()
            futs = [ex.submit(job, b) for b in batches[:workers + 1]]
            nxt = workers + 1
            keep = []
            for i in range(len(batches)):
                wb, ev = futs[i].result()
                if nxt < len(batches):
                    futs.append(ex.submit(job, batches[nxt])); nxt += 1
                if do_step:
                    torch.cuda.current_stream().wait_event(ev)
                    loss = model.run_loss(wb); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
                keep.append(wb); keep = keep[-4:]
            torch.cuda.synchronize()
            print("workers %d, %s: %.2f ms per batch" % (workers, "train step" if do_step else "prepare only", 1e3 * (time.perf_counter() - t0) / len(batches)))
