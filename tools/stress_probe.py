import os, sys, itertools
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from temp_amd.bi_dynamic_rgcn import BiDynamicRGCN
from temp_amd.dynamic_rgcn import DynamicRGCN
from tests.window_cases import make_args, slice_snapshots
DEV = torch.device("cuda:0")
s = slice_snapshots()
bad = 0
for module, rec_only, type1, score, D, B in itertools.product(("GRRGCN", "BiGRRGCN"), (True, False), (False, True), ("complex", "distmult"), (32,), (8,)):
    args = make_args(module=module, rec_only_last_layer=rec_only, type1=type1, embed_size=D, hidden_size=D, n_bases=B, train_seq_len=6,
                     test_seq_len=6, score_function=score, negative_rate=30, num_pos_facts=25)
    cls = BiDynamicRGCN if module.startswith("Bi") else DynamicRGCN
    torch.manual_seed(11)
    m = cls(args, s["num_e"], s["num_r"], s["tr"], s["va"], s["te"]).to(DEV)
    t_list = torch.tensor([s["times"][i] for i in (21, 14, 9, 4, 1)])
    m.sample_rng = np.random.default_rng(5)
    wb = m.prepare(t_list, 6, train=True)
    m.seed_rng = np.random.default_rng(9)
    samples = m._samples_from_plan(wb)
    res = []
    for fused in (True, False):
        m.fused_loss = fused
        for p in m.parameters():
            p.grad = None
        loss = m.run_loss(wb, samples)
        loss.backward()
        res.append((loss.item(), m.ent_embeds.grad.clone(), m.rel_embeds.grad.clone(), m.ent_encoder.layer_1.weight.grad.clone()))
    m.fused_loss = True
    ok = abs(res[0][0] - res[1][0]) < 3e-5 * abs(res[1][0])
    for a, b in zip(res[0][1:], res[1][1:]):
        ok = ok and torch.allclose(a, b, rtol=2e-4, atol=2e-5 * float(b.abs().max()) + 1e-7)
    ranks, l = m.evaluate(t_list, val=True)
    print("%-9s rec_only=%-5s type1=%-5s %-8s fused-all=%s loss %.5f vs %.5f  eval ranks %d  %s" % (module, rec_only, type1, score, m._fused_all_entity_ok(wb), res[0][0], res[1][0], ranks.numel(), "ok" if ok else "MISMATCH"))
    bad += 0 if ok else 1
print("mismatches:", bad)
