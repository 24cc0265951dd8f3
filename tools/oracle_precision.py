"""How noisy is the fp32 CPU oracle itself at BASELINE size?  Runs 3 full S-gdelt windows (fwd+bwd) through oracle/temp_oracle.py in
fp32 and in fp64 and prints the differences (result quoted in tests/test_gpu_parity_r2.py)."""
import sys, torch, numpy as np, time
sys.path.insert(0, "/root/repo")
from oracle import temp_oracle as O
from temp_amd import synthetic
import tests.test_gpu_parity_r2 as R
w = synthetic.workload("S-gdelt", seed=0)
cfg = dict(module="BiGRRGCN", n_bases=100, inv_temperature=0.1, rec_only_last_layer=True, use_time_embedding=False)
targets = sorted(synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 0)[:3], reverse=True)
L, D = w["L"], w["D"]
res = {}
for dt in (torch.float32, torch.float64):
    om = O.init_model(cfg, w["num_ents"], w["num_rels"], w["num_times"], D, seed=1)
    om = O.map_params(om, lambda t: t.to(dt))
    gd = {t: O.SnapGraph(g.n, g.src, g.dst, g.rel, g.gids) for t, g in w["snapshots"].items()}
    times = sorted(gd.keys())
    leaves = O.leaf_tensors(om)
    for v in leaves.values(): v.requires_grad_(True)
    t0 = time.time()
    tf, tb = O.get_batch_graph_list_bi(targets, L, times)
    Hf = O.bi_pre_forward(om, cfg, gd, tf, L, True)
    Hb = O.bi_pre_forward(om, cfg, gd, tb, L, False)
    want = O.bi_target_embeds(om, cfg, Hf, Hb, [gd[t] for t in targets], tf[-1], L)
    ups = R._upstream([x.shape[0] for x in want], D, 7)
    sum((p * u.to(dt)).sum() for p, u in zip(want, ups)).backward()
    print(dt, time.time() - t0)
    res[dt] = (torch.cat(want).detach().double(), om["ent_embeds"].grad.double(), om["ent_encoder"]["layer_2"]["forward_rnn"][0]["w_hh"].grad.double())
a, b = res[torch.float32], res[torch.float64]
for nm, x, y in zip(("out", "d_ent", "d_whh"), a, b):
    err = (x - y).abs()
    print(nm, "max abs err %.3e  max|ref| %.3e  rel-to-max %.3e ; worst rel elementwise(>1e-3 of max) %.3e" % (err.max(), y.abs().max(), err.max() / y.abs().max(), (err / y.abs().clamp_min(1e-3 * y.abs().max())).max()))
