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
m.use_rec_stack = "--loop" not in sys.argv          # --loop: the reference-granular position loop (one RRGCN.forward per position)
print("path:", "one-node position loop (rec_stack.py)" if m.use_rec_stack else "reference-granular loop")
m.sample_rng = np.random.default_rng(2)
m.corrupter = CorruptTriples(m.args, w["snapshots"], seed=5)
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
batches = [synthetic.default_targets(w["num_times"], w["L"], w["bsz"], r) for r in range(40)]
for b in batches[:5]:
    loss = m(torch.tensor(b)); opt.zero_grad(); loss.backward(); opt.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for b in batches[5:]:
    loss = m(torch.tensor(b)); opt.zero_grad(); loss.backward(); opt.step()
torch.cuda.synchronize()
print("generic path (rec_only_last_layer=False), S-icews14: %.2f ms/step, loss %.3f" % (1e3 * (time.perf_counter() - t0) / 35, loss.item()))
tl = bench.train_loop(m, w, 40)
print("train loop (bench.train_loop: new batch every step, Adam): inline prepare %.2f ms/step, background prefetcher %.2f ms/step"
      % (tl["inline_prepare_ms_per_step"], tl["prefetcher_ms_per_step"]))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for b in batches[5:15]:
    loss = m(torch.tensor(b)); opt.zero_grad(); loss.backward(); opt.step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(38)
