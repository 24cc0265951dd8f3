"""Diagnostic (GPU box): device-side subsample inside a batched training step vs host-built subgraphs of the same edges."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from temp_amd import synthetic, functional as TF, snapshot as S
DEV = torch.device("cuda:0")
w = synthetic.workload("S-gdelt", seed=0)
model = bench.build_model(w, DEV)
R2 = 2 * w["num_rels"]
graphs = [w["snapshots"][t] for t in (100, 50, 20)]
subs = S.device_subsample(graphs, [g.number_of_edges() // 2 for g in graphs], [5, 6, 7], DEV, R2, want_mask=True)
torch.cuda.synchronize()
gen = torch.Generator().manual_seed(1)
D, B = w["D"], w["B"]
wgt = (torch.randn(R2, B * (D // B) ** 2, generator=gen) * 0.3).to(DEV)
lw = (torch.randn(D, D, generator=gen) * 0.1).to(DEV)
for i, (g, sub) in enumerate(zip(graphs, subs)):
    idx = sub.edge_ids
    print("graph", i, "kept", len(idx), "of", g.number_of_edges())
    host = g.edge_subgraph(idx)
    h = torch.randn(g.n, D, generator=gen).to(DEV)
    ya = TF.rgcn_layer(h, sub.device_graph(DEV, R2), wgt, lw, None, B, None)
    yb = TF.rgcn_layer(h, host.device_graph(DEV, R2), wgt, lw, None, B, None)
    print("  single-graph layer max err", float((ya - yb).abs().max()), "max", float(yb.abs().max()))
# union of [history graph, sub...] through the assembler
hist = [w["snapshots"][t] for t in range(30, 60)]
for name, members in (("device", hist + subs), ("host", hist + [g.edge_subgraph(s.edge_ids) for g, s in zip(graphs, subs)])):
    u = S.batch(members)
    dg = u.device_graph(DEV, R2)
    gen2 = torch.Generator().manual_seed(2)
    h = torch.randn(u.n, D, generator=gen2).to(DEV)
    y = TF.rgcn_layer(h, dg, wgt, lw, None, B, None)
    print(name, "union n", u.n, "E", u.number_of_edges(), "sum", float(y.double().sum()), "abs", float(y.double().abs().sum()))
    if name == "device":
        yd = y
    else:
        err = (yd - y).abs()
        print("union layer max err", float(err.max()), "rows with err>1e-4:", int((err.max(1).values > 1e-4).sum()), "first bad row", int(torch.nonzero(err.max(1).values > 1e-4)[0]) if (err.max(1).values > 1e-4).any() else -1, "target rows start", sum(g.n for g in hist))
