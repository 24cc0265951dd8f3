import os, sys, faulthandler
faulthandler.enable()
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
import bench
from temp_amd import synthetic
from temp_amd.dist import ShardedStep, SnapshotShardedEncoder
variant = sys.argv[1]
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29577"
dist.init_process_group("nccl", rank=0, world_size=1)
DEV = torch.device("cuda:0")
w = synthetic.workload("S-gdelt", seed=0)
model = bench.build_model(w, DEV)
nb = 8 if "b8" in variant else 3
targets = synthetic.default_targets(w["num_times"], w["L"], nb, 0)
params = [p for p in model.parameters()]
if "ref" in variant:
    model.sample_rng = np.random.default_rng(2)
    wb = model.prepare(targets, w["L"], train=True)
    ref_out = model.run(wb)[0]
    ref_out.backward(torch.ones_like(ref_out))
    for p in params:
        p.grad = None
model.sample_rng = np.random.default_rng(2)
enc = SnapshotShardedEncoder(model)
sb = enc.prepare(targets, w["L"], train=True)
if "eager" in variant:
    st = ShardedStep(enc, sb, params, graphs=False, average=True, force_allreduce=True)
    st.step(); st.step()
    torch.cuda.synchronize()
    for p in params:
        p.grad = None
print("capturing", flush=True)
st = ShardedStep(enc, sb, params, graphs=True, average=True, force_allreduce=True)
st.step(); st.step()
torch.cuda.synchronize()
print("variant", variant, "ok", flush=True)
dist.destroy_process_group()
