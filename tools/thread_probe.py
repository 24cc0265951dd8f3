#!/usr/bin/env python3
"""prepare() cost: main thread vs worker thread, default stream vs side stream (development probe)."""
import os, sys, time, threading
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from temp_amd import synthetic
w = synthetic.workload("S-gdelt", seed=0)
dev = torch.device("cuda:0")
model = bench.build_model(w, dev)
model.sample_rng = np.random.default_rng(2)
batches = [synthetic.default_targets(w["num_times"], w["L"], w["bsz"], r) for r in range(30)]
def run(side):
    st = torch.cuda.Stream(dev) if side else None
    ts = []
    for b in batches:
        torch.cuda.synchronize()
        t = time.perf_counter()
        if st is None:
            model.prepare(b, w["L"], True)
        else:
            with torch.cuda.stream(st):
                model.prepare(b, w["L"], True)
        ts.append(time.perf_counter() - t)
    return 1e3 * float(np.mean(ts[5:]))
print("main thread, default stream: %.2f ms" % run(False))
print("main thread, side stream   : %.2f ms" % run(True))
out = {}
for side in (False, True):
    th = threading.Thread(target=lambda: out.__setitem__(side, run(side)))
    th.start(); th.join()
    print("worker thread, %s stream: %.2f ms" % ("side" if side else "default", out[side]))
