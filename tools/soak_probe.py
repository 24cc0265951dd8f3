import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from temp_amd import synthetic
from temp_amd.sampling import CorruptTriples
w = synthetic.workload("S-gdelt", seed=0)
dev = torch.device("cuda:0")
model = bench.build_model(w, dev)
model.sample_rng = np.random.default_rng(2)
model.corrupter = CorruptTriples(model.args, w["snapshots"], seed=5)
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
import resource
t0 = time.perf_counter()
workers = int(os.environ.get("PROBE_WORKERS", "0"))          # 0: prepare inline; N: temp_amd.prefetch.BatchPrefetcher with N workers
steps = int(os.environ.get("PROBE_STEPS", "3000"))
batches = (synthetic.default_targets(w["num_times"], w["L"], w["bsz"], i) for i in range(steps))
if workers:
    from temp_amd.prefetch import BatchPrefetcher
    source = BatchPrefetcher(model, batches, seq_len=w["L"], depth=2, workers=workers, batch_seeds=True)
else:
    source = (model.prepare(b, w["L"], True) for b in batches)
for i, wb in enumerate(source):
    loss = model.run_loss(wb); opt.zero_grad(set_to_none=True); loss.backward(); opt.step()
    if i % 500 == 499:
        torch.cuda.synchronize()
        print("step %d: %.2f ms/step, loss %.3f, gpu alloc %.0f MB (max %.0f), reserved %.0f MB, host rss %.0f MB" % (
            i + 1, 1e3 * (time.perf_counter() - t0) / (i + 1), loss.item(), torch.cuda.memory_allocated() / 2**20,
            torch.cuda.max_memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024), flush=True)
