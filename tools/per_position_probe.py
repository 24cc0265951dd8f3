import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch, bench
from temp_amd import _lib, synthetic
from temp_amd import gru_chain as GC
lib = _lib.load()
dev = torch.device("cuda", 0)
w = synthetic.workload("S-gdelt", seed=0)
model = bench.build_model(w, dev)
model.sample_rng = np.random.default_rng(2)
GC.CHAIN_KERNELS = False
wb = model.prepare(synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 0), w["L"], train=True)
st = bench.GraphStep(lambda: model.run(wb)[0], list(model.parameters()), graph=False)
for _ in range(3): st.eager()
torch.cuda.synchronize()
tr = bench.traced_steps(st.eager, 3, lib)
for k, v in sorted(tr.items(), key=lambda kv: -kv[1]["ms_per_step"])[:12]:
    print("%-28s launches %5.1f  ms/step %7.3f  avg us %7.1f" % (k, v["launches_per_step"], v["ms_per_step"], 1e3 * v["avg_ms"]))
