#!/usr/bin/env python3
"""The snapshot-sharded step (temp_amd.dist.ShardedStep) on ONE RCCL rank, alone in a process: the thing to put under
rocprofv3 --kernel-trace when looking for what `extra.sharded_1rank` pays over the unsharded step.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_shard1 -o t -- python tools/shard_trace.py --steps 30
    python tools/step_sequence.py gpurun_out/prof_shard1/t_kernel_trace.csv"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from temp_amd import synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--no-graph", action="store_true")
    a = ap.parse_args()
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    device = torch.device("cuda", 0)
    w = synthetic.workload("S-gdelt", seed=0)
    model = bench.build_model(w, device, "gru")
    from temp_amd.dist import ShardedStep, SnapshotShardedEncoder
    targets = synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 0)
    model.sample_rng = np.random.default_rng(2)
    enc = SnapshotShardedEncoder(model)
    sb = enc.prepare(targets, w["L"], train=True)
    st = ShardedStep(enc, sb, [p for p in model.parameters()], graphs=not a.no_graph, average=False, force_allreduce=True)
    for _ in range(5):
        st.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        st.step()
    torch.cuda.synchronize()
    print("sharded step on one rank: %.3f ms" % (1e3 * (time.perf_counter() - t0) / a.steps))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
