"""Diagnostic (GPU box): gradient error of the self-attention encoder step vs the fp64 oracle, per parameter."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import temp_oracle as O
import bench
from temp_amd import synthetic
import tests.test_gpu_parity_r2 as R

DEV = torch.device("cuda:0")
w = synthetic.workload("S-gdelt", seed=0)
model = bench.build_model(w, DEV, "attention")
targets = sorted(synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 0)[:2], reverse=True)
L, D = w["L"], w["D"]
per_graph, wb, tables = model.encode(torch.tensor(targets), L, train=False)
ups = R._upstream([p.shape[0] for p in per_graph], D, 11)
sum((p * u.to(DEV)).sum() for p, u in zip(per_graph, ups)).backward()
torch.cuda.synchronize()
om, cfg, gd = R._oracle_model(model, w, "BiSARGCN", te=True)
cfg["learnable_lambda"] = False
times = sorted(gd.keys())
leaves = O.leaf_tensors(om)
for v in leaves.values():
    v.requires_grad_(True)
want, *_ = O.sa_encode(om, cfg, gd, targets, times, L, [gd[t] for t in targets], bi=True)
sum((p * u.double()).sum() for p, u in zip(want, ups)).backward()
sd = dict(model.named_parameters())
for k, v in leaves.items():
    if v.grad is None:
        continue
    parts = k.split(".")
    name = k + ".weight" if parts[-1] in ("q_linear", "k_linear", "v_linear") else k
    if name not in sd or sd[name].grad is None:
        print("skip", k); continue
    g = sd[name].grad.detach().cpu().double()
    err = (g - v.grad).abs()
    print("%-40s max|ref| %.3e  max err %.3e  rel-to-max %.2e  n(err>1e-4*max) %d / %d" % (k, v.grad.abs().max(), err.max(), err.max() / v.grad.abs().max(), int((err > 1e-4 * v.grad.abs().max()).sum()), err.numel()))
for i, (a, b) in enumerate(zip(per_graph, want)):
    print("out", i, float((a.detach().cpu().double() - b).abs().max()))
