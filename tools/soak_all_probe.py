import os, sys, time, resource
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from temp_amd import synthetic
from temp_amd.sampling import CorruptTriples
for wl, enc in (("S-icews0515", "gru"), ("S-gdelt", "attention"), ("S-icews14", "gru")):
    w = synthetic.workload(wl, seed=0)
    dev = torch.device("cuda:0")
    model = bench.build_model(w, dev, enc)
    model.sample_rng = np.random.default_rng(2)
    model.corrupter = CorruptTriples(model.args, w["snapshots"], seed=5)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    t0 = time.perf_counter(); first = None
    for i in range(1200):
        b = synthetic.default_targets(w["num_times"], w["L"], w["bsz"], i)
        wb = model.prepare(b, w["L"], True)
        loss = model.run_loss(wb); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
        if i == 99: first = loss.item()
    torch.cuda.synchronize()
    print("%-12s %-9s 1200 steps: %.2f ms/step, loss %.3f -> %.3f, gpu alloc %.0f MB (reserved %.0f), host rss %.0f MB" % (
        wl, enc, 1e3 * (time.perf_counter() - t0) / 1200, first, loss.item(), torch.cuda.memory_allocated() / 2**20,
        torch.cuda.memory_reserved() / 2**20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024), flush=True)
    del model, opt
    torch.cuda.empty_cache()
