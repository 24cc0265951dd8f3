#!/usr/bin/env python3
"""Training-step wall time of the static RGCN baseline (BASELINE config 1 shape) on S-icews14 (development probe)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from temp_amd import synthetic
from temp_amd.static_rgcn import StaticRGCN
w = synthetic.workload("S-icews14", seed=0)
dev = torch.device("cuda:0")
args = bench.make_args(w, "SRGCN")
torch.manual_seed(1)
m = StaticRGCN(args, w["num_ents"], w["num_rels"], w["snapshots"], w["snapshots"], w["snapshots"]).to(dev)
from temp_amd.sampling import CorruptTriples
m.corrupter = CorruptTriples(m.args, w["snapshots"], seed=5)
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
batches = [synthetic.default_targets(w["num_times"], w["L"], w["bsz"], r) for r in range(45)]
for b in batches[:5]:
    loss = m(torch.tensor(b)); opt.zero_grad(); loss.backward(); opt.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for b in batches[5:]:
    loss = m(torch.tensor(b)); opt.zero_grad(); loss.backward(); opt.step()
torch.cuda.synchronize()
print("static RGCN, S-icews14: %.2f ms/step, loss %.3f" % (1e3 * (time.perf_counter() - t0) / 40, loss.item()))
