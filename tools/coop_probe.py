#!/usr/bin/env python3
"""Training loop (new batch every step, bench.train_loop's loop) with the prefetcher's cooperative gate on / off and 1 / 2 workers.
python tools/coop_probe.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from temp_amd import synthetic  # noqa: E402
from temp_amd.prefetch import BatchPrefetcher  # noqa: E402
from temp_amd.sampling import CorruptTriples  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
w = synthetic.workload("S-gdelt", seed=0)
dev = torch.device("cuda:0")
model = bench.build_model(w, dev)
model.sample_rng = np.random.default_rng(2)
model.corrupter = CorruptTriples(model.args, w["snapshots"], seed=5)
opt = model.configure_optimizers()
WARM = 10
batches = [synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 1000 + r) for r in range(steps + WARM)]
for b in batches:
    model.prepare(b, w["L"], True)


import threading
_prep = {"t": 0.0, "n": 0}
_plock = threading.Lock()
_orig_prepare = model.prepare


def _timed_prepare(*a, **k):
    t = time.perf_counter()
    r = _orig_prepare(*a, **k)
    dt = time.perf_counter() - t
    with _plock:
        _prep["t"] += dt
        _prep["n"] += 1
    return r


model.prepare = _timed_prepare
last = {}


def timed(source):
    t0 = None
    it = iter(source)
    i = 0
    t_wait = t_issue = 0.0
    while True:
        ta = time.perf_counter()
        try:
            wb = next(it)
        except StopIteration:
            break
        tb = time.perf_counter()
        if i == WARM:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            t_wait = t_issue = 0.0
            _prep["t"], _prep["n"] = 0.0, 0
            tb = t0
            ta = t0
        loss = model.run_loss(wb)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        tc = time.perf_counter()
        t_wait += tb - ta
        t_issue += tc - tb
        i += 1
    torch.cuda.synchronize()
    last.update(wait=1e3 * t_wait / steps, issue=1e3 * t_issue / steps, prep=1e3 * _prep["t"] / max(_prep["n"], 1))
    return 1e3 * (time.perf_counter() - t0) / steps


def fmt():
    return "(main: next() %.2f ms, issue %.2f ms; prepare %.2f ms per call)" % (last["wait"], last["issue"], last["prep"])


wbs = [model.prepare(b, w["L"], True) for b in batches[:40]]
print("resident batches (device + issue only): %.2f ms/step" % timed(wbs[i % 40] for i in range(steps + WARM)), fmt())
print("inline prepare: %.2f ms/step" % timed(model.prepare(b, w["L"], True) for b in batches), fmt())
for coop, workers, depth in ((True, 2, 2), (True, 3, 2), (True, 4, 2), (True, 2, 4), (True, 3, 4), (False, 2, 2), (True, 2, 2)):
    if True:
        if True:
            ms = timed(BatchPrefetcher(model, batches, seq_len=w["L"], depth=depth, workers=workers, batch_seeds=True, cooperative=coop))
            print("prefetcher cooperative=%d workers=%d depth=%d: %.2f ms/step" % (coop, workers, depth, ms), fmt())
