#!/usr/bin/env python3
"""How the CPU oracle scales over PROCESSES on this host: bench.cpu_all_cores_parallel with 1 .. 16 processes of 16 (or 8) torch
threads.  python tools/cpu_scale_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from temp_amd import synthetic  # noqa: E402

w = synthetic.workload("S-gdelt", seed=0)
print("CPUs this process may use at once (cgroup quota / affinity): %d of %d hardware threads" % (bench._cpu_quota(), os.cpu_count()))
for procs, thr in ((1, 16), (2, 16), (4, 16), (8, 16), (8, 8), (16, 8)):
    r = bench.cpu_all_cores_parallel(w, threads_per_process=thr, seconds=12.0, timeout_s=60.0, processes=procs)
    print("%2d processes x %2d threads: %s edges/s, %s batches, median batch %s s" % (procs, thr, r.get("value"), r.get("batches"), r.get("median_batch_s")), flush=True)
