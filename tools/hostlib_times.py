#!/usr/bin/env python3
"""Wall time per batch inside every call of the C++ planner library during BiDynamicRGCN.prepare (development probe)."""
import os, sys, time, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from temp_amd import synthetic, _hostlib
acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
lib = _hostlib.load()
class Timed:
    def __init__(self, name, fn): self.name, self.fn = name, fn
    def __call__(self, *a):
        t = time.perf_counter(); r = self.fn(*a); acc[self.name] += time.perf_counter() - t; cnt[self.name] += 1; return r
class Proxy:
    def __getattr__(self, k): return Timed(k, getattr(lib, k))
_hostlib._lib = Proxy()
_hostlib.load = lambda: _hostlib._lib
w = synthetic.workload("S-gdelt", seed=0)
dev = torch.device("cuda:0")
model = bench.build_model(w, dev)
model.sample_rng = np.random.default_rng(2)
batches = [synthetic.default_targets(w["num_times"], w["L"], w["bsz"], rep) for rep in range(20)]
for b in batches: model.prepare(b, w["L"], train=True)
acc.clear(); cnt.clear()
torch.cuda.synchronize(); t0 = time.perf_counter()
for b in batches: model.prepare(b, w["L"], train=True)
torch.cuda.synchronize(); tot = time.perf_counter() - t0
print("prepare %.2f ms per batch; inside the planner library %.2f ms" % (1e3 * tot / 20, 1e3 * sum(acc.values()) / 20))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-28s %.3f ms per batch in %.1f calls" % (k, 1e3 * v / 20, cnt[k] / 20))
print("num_pos_facts", model.args.num_pos_facts, "negative_rate", model.args.negative_rate)
