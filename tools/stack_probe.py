"""GPU box: where the default-flags step (rec_only_last_layer=False, S-icews14) spends its time -- host issue time against GPU
time of the encoder + loss step on ONE resident batch, for the one-node position loop (rec_stack.py) and, with --loop, the
reference-granular loop; then the kernel list of one step (in-library event trace)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from temp_amd import synthetic
from temp_amd.dynamic_rgcn import DynamicRGCN
from temp_amd.sampling import CorruptTriples
w = synthetic.workload("S-icews14", seed=0)
dev = torch.device("cuda:0")
args = bench.make_args(w, "GRRGCN"); args.rec_only_last_layer = False
torch.manual_seed(1)
m = DynamicRGCN(args, w["num_ents"], w["num_rels"], w["snapshots"], w["snapshots"], w["snapshots"]).to(dev)
m.use_rec_stack = "--loop" not in sys.argv
m.sample_rng = np.random.default_rng(2)
m.corrupter = CorruptTriples(m.args, w["snapshots"], seed=5)
b = synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 3)
wb = m.prepare(b, w["L"], True)
print("path:", "one-node loop" if m.use_rec_stack else "reference-granular loop", " positions", len(wb.steps), " rows", wb.n_node_visits, " edge visits", wb.n_edge_visits)
def step():
    loss = m.run_loss(wb)
    for p in m.parameters():
        p.grad = None
    loss.backward()
for _ in range(5):
    step()
torch.cuda.synchronize()
for name in ("encoder+loss fwd+bwd",):
    host, tot = [], []
    for _ in range(20):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        step()
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        host.append(t1 - t0); tot.append(t2 - t0)
    print("%s: host issue %.2f ms, until the GPU is done %.2f ms (median of 20)" % (name, 1e3 * np.median(host), 1e3 * np.median(tot)))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
# GPU-only time: queue 5 steps back to back, events around them
torch.cuda.synchronize(); e0.record()
for _ in range(5):
    step()
e1.record(); torch.cuda.synchronize()
print("5 steps queued back to back: %.2f ms per step (max of host issue and GPU time)" % (e0.elapsed_time(e1) / 5))
from temp_amd import _lib
tr = bench.traced_steps(step, 1, _lib.load())
tot = sum(v["ms_per_step"] for v in tr.values())
print("event trace of one step: %d launches, %.2f ms of kernel time" % (sum(v["launches_per_step"] for v in tr.values()), tot))
for k, v in sorted(tr.items(), key=lambda kv: -kv[1]["ms_per_step"])[:16]:
    print("  %-34s x%-4d %.3f ms" % (k, v["launches_per_step"], v["ms_per_step"]))
