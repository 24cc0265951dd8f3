#!/usr/bin/env python3
"""Un-profiled wall time of the stages of BiDynamicRGCN.prepare (steady state), by wrapping the stage functions with timers."""
import os, sys, time, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from temp_amd import synthetic, window, snapshot, functional, sampling, gru_chain, dynamic_rgcn, bi_dynamic_rgcn
acc = collections.defaultdict(float)
def wrap(mod, name, label=None):
    f = getattr(mod, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[label or name] += time.perf_counter() - t; return r
    setattr(mod, name, g)
wrap(window, "ChainPlan"); wrap(bi_dynamic_rgcn, "ChainPlan", "ChainPlan"); wrap(dynamic_rgcn, "ChainPlan", "ChainPlan")
wrap(dynamic_rgcn, "concat_steps_dedup"); wrap(snapshot, "union_graph_packed"); wrap(snapshot, "build_view")
wrap(functional, "gather_inverse"); wrap(gru_chain.GruProgram, "chain_plan", "program.chain_plan"); wrap(gru_chain.GruProgram, "chain_tables", "program.chain_tables"); wrap(sampling, "plan_batch_loss"); wrap(gru_chain.GruProgram, "upload", "program.upload")
wrap(dynamic_rgcn.DynamicRGCN, "sample_target_graphs"); wrap(bi_dynamic_rgcn.BiDynamicRGCN, "_build_program"); wrap(bi_dynamic_rgcn.BiDynamicRGCN, "_bi_target")
wrap(dynamic_rgcn.DynamicRGCN, "_all_maps"); wrap(dynamic_rgcn.DynamicRGCN, "_upload"); wrap(dynamic_rgcn.DynamicRGCN, "_plan_loss")
w = synthetic.workload(sys.argv[1] if len(sys.argv) > 1 else "S-gdelt", seed=0)
dev = torch.device("cuda:0")
model = bench.build_model(w, dev)
model.sample_rng = np.random.default_rng(2)
batches = [synthetic.default_targets(w["num_times"], w["L"], w["bsz"], rep) for rep in range(20)]
for b in batches: model.prepare(b, w["L"], train=True)
acc.clear()
torch.cuda.synchronize(); t0 = time.perf_counter()
for b in batches: model.prepare(b, w["L"], train=True)
torch.cuda.synchronize(); tot = time.perf_counter() - t0
print("prepare %.2f ms per batch" % (1e3 * tot / len(batches)))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-24s %.2f ms" % (k, 1e3 * v / len(batches)))
