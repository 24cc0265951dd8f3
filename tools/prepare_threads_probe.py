#!/usr/bin/env python3
"""Throughput of `prepare` alone (no training step beside it): k threads, each on its own HIP stream, with and without the planning
token of temp_amd._lib -- how much of a batch's planning can overlap another's.  python tools/prepare_threads_probe.py"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from temp_amd import _lib, synthetic  # noqa: E402
from temp_amd.sampling import CorruptTriples  # noqa: E402
from temp_amd.tkg_module import TKG_Module  # noqa: E402

w = synthetic.workload("S-gdelt", seed=0)
dev = torch.device("cuda:0")
model = bench.build_model(w, dev)
model.sample_rng = np.random.default_rng(2)
model.corrupter = CorruptTriples(model.args, w["snapshots"], seed=5)
N = 120
batches = [synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 1000 + r) for r in range(N)]
for b in batches:
    model.prepare(b, w["L"], True)
torch.cuda.synchronize()
sys.setswitchinterval(float(os.environ.get("TEMP_SWITCH_INTERVAL", "5e-5")))
from temp_amd import _hostlib  # noqa: E402
calls = _hostlib.load()
acc = {}
alock = threading.Lock()
for name in _hostlib.SYMBOLS:
    if name == "temp_host_abi_version":
        continue

    def mk(fn, name):
        def call(*a):
            t = time.perf_counter()
            try:
                return fn(*a)
            finally:
                d = time.perf_counter() - t
                with alock:
                    e = acc.setdefault(name, [0, 0.0])
                    e[0] += 1
                    e[1] += d
        return call
    setattr(calls, name, mk(getattr(calls, name), name))


def run(k, token):
    gate = threading.Event()
    gate.set()
    nxt = [0]
    lock = threading.Lock()

    def work():
        st = torch.cuda.Stream(dev)
        while True:
            with lock:
                i = nxt[0]
                nxt[0] += 1
            if i >= N:
                break
            TKG_Module._rng_override.rng = np.random.default_rng(i)
            if token:
                _lib.coop_begin(gate)
            try:
                with torch.cuda.stream(st):
                    model.prepare(batches[i], w["L"], True)
            finally:
                _lib.coop_end()
        st.synchronize()

    ths = [threading.Thread(target=work) for _ in range(k)]
    t0 = time.perf_counter()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return 1e3 * (time.perf_counter() - t0) / N


for k in (1, 2):
    for token in (False, True):
        acc.clear()
        ms = run(k, token)
        tot = sum(v[1] for v in acc.values())
        print("threads %d token %d: %.2f ms per batch; planner calls %.2f ms per batch (wall seen from Python: %s)" % (
            k, token, ms, 1e3 * tot / N, ", ".join("%s %d x %.0f us" % (n[10:], v[0] // N, 1e6 * v[1] / max(v[0], 1)) for n, v in sorted(acc.items()))))
