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
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for b in batches[5:15]:
    loss = m(torch.tensor(b)); opt.zero_grad(); loss.backward(); opt.step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(38)
