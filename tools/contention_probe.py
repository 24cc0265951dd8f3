#!/usr/bin/env python3
"""What inflates the step's issue code beside a prefetch worker: the interpreter lock or the HIP runtime's own locks?  The issue code
of a prepared batch (run_loss + backward + Adam) timed alone, beside a pure-Python thread, beside a thread that only enqueues small
pinned-host -> device copies on its own stream, and beside a numpy-in-C thread (lock released)."""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from temp_amd import synthetic  # noqa: E402
from temp_amd.sampling import CorruptTriples  # noqa: E402

w = synthetic.workload("S-gdelt", seed=0)
dev = torch.device("cuda:0")
model = bench.build_model(w, dev)
model.sample_rng = np.random.default_rng(2)
model.corrupter = CorruptTriples(model.args, w["snapshots"], seed=5)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
wbs = [model.prepare(synthetic.default_targets(w["num_times"], w["L"], w["bsz"], r), w["L"], True) for r in range(24)]


def step(wb):
    loss = model.run_loss(wb)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


def measure(label, target):
    stop = threading.Event()
    th = threading.Thread(target=target, args=(stop,)) if target else None
    if th:
        th.start()
    for wb in wbs[:4]:
        step(wb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for wb in wbs[4:]:
        step(wb)
    ti = time.perf_counter() - t0
    torch.cuda.synchronize()
    ta = time.perf_counter() - t0
    stop.set()
    if th:
        th.join()
    print("%-46s issue %.2f ms/step, wall %.2f ms/step" % (label, 1e3 * ti / 20, 1e3 * ta / 20))


def py_busy(stop):
    x = 0
    while not stop.is_set():
        for i in range(2000):
            x += i * i


def copy_busy(stop):
    s = torch.cuda.Stream()
    src = torch.empty(65536, dtype=torch.int32).pin_memory()
    dst = torch.empty(65536, dtype=torch.int32, device=dev)
    with torch.cuda.stream(s):
        while not stop.is_set():
            for _ in range(14):
                dst.copy_(src, non_blocking=True)
            time.sleep(0.002)


def numpy_busy(stop):
    a = np.random.rand(400, 400)
    while not stop.is_set():
        a @ a


import sys as _s
measure("alone", None)
measure("beside a pure-Python thread", py_busy)
_s.setswitchinterval(5e-5)
measure("beside a pure-Python thread, switch 50 us", py_busy)
measure("beside 14 small async copies every 2 ms", copy_busy)
measure("beside numpy matmuls (lock released)", numpy_busy)
