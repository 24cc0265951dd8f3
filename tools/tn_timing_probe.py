"""GPU box: why bench.py's HIP-event average of the dominant kernel (k_gemm_tn_bx8, traced in EAGER steps) sits 7-10 % above
rocprofv3's average of the same kernel (mostly HIP-graph replays): the same product timed (a) inside eager steps, (b) in 40
back-to-back launches, (c) in 40 launches with 2 ms of host idle before each."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from temp_amd import _lib, synthetic, backend as TB
lib = _lib.load()
dev = torch.device("cuda", 0)
w = synthetic.workload("S-gdelt", seed=0)
model = bench.build_model(w, dev)
model.sample_rng = np.random.default_rng(2)
wb = model.prepare(synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 0), w["L"], train=True)
st = bench.GraphStep(lambda: model.run(wb)[0], list(model.parameters()), graph=False)
for _ in range(3):
    st.eager()
torch.cuda.synchronize()
tr = bench.traced_steps(st.eager, 5, lib)
print("(a) eager steps:            k_gemm_tn_bx8 %.1f us  (x%d per step)" % (1e3 * tr["k_gemm_tn_bx8"]["avg_ms"], tr["k_gemm_tn_bx8"]["launches_per_step"]))
be = TB.get_backend()
M = 60000
a, b = torch.randn(M, 600, device=dev), torch.randn(M, 200, device=dev)
out = torch.empty(600, 200, device=dev)
for _ in range(3):
    be.linear_tn(a, b, out)
torch.cuda.synchronize()
def many(idle):
    for _ in range(40):
        if idle:
            torch.cuda.synchronize(); time.sleep(idle)
        be.linear_tn(a, b, out)
    torch.cuda.synchronize()
for idle, what in ((0.0, "(b) 40 back to back:       "), (0.002, "(c) 2 ms idle before each: "), (0.0, "(b) again:                 ")):
    tr = bench.traced_steps(lambda: many(idle), 1, lib)
    print("%s k_gemm_tn_bx8 %.1f us" % (what, 1e3 * tr["k_gemm_tn_bx8"]["avg_ms"]))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); many(0.0); e1.record(); torch.cuda.synchronize()
print("(d) one event pair around 40 launches (TN + its slice reduction): %.1f us per launch" % (1e3 * e0.elapsed_time(e1) / 40))
