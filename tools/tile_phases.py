#!/usr/bin/env python3
"""Development probe: phase stamps of the LDS-tiled aggregation kernel (csrc/rgcn_tile.hpp) on the S-gdelt batch of the bench."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from temp_amd import _lib, synthetic
lib = _lib.load()
w = synthetic.workload("S-gdelt", seed=0)
dev = torch.device("cuda:0")
model = bench.build_model(w, dev)
model.sample_rng = np.random.default_rng(2)
wb = model.prepare(synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 0), w["L"], train=True)
enc = model.ent_encoder
dg = wb.g_all.device_graph(dev, 2 * w["num_rels"])
print("members", dg.c.members.n_members, "max nodes/edges/chunks", dg.c.members.max_nodes, dg.c.members.max_edges, list(dg.c.members.max_chunks))
y1 = enc.layer_1.conv_table(wb.g_all, model.ent_embeds, wb.ids_all, wb.ids_inv)
words = 8 * 4096
buf = torch.zeros(words, dtype=torch.int64, device=dev)
for rep, var in enumerate([0, 0, 1, 2, 3, 4, 7, 8]):
    lib.temp_set_option(_lib.OPT_DEBUG, var)
    buf.zero_()
    lib.temp_set_debug_buffer(buf.data_ptr(), words)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    y2 = enc.layer_2.conv(wb.g_all, y1)
    ev1.record()
    torch.cuda.synchronize()
    lib.temp_set_debug_buffer(None, 0)
    st = buf.cpu().numpy().reshape(-1, 8)
    st = st[st[:, 0] != 0]
    d = np.diff(st, axis=1).astype(np.float64)
    names = ["issue loads", "barrier", "sort", "commit", "barrier", "main", "tail-barrier"]
    print("VAR %d" % var, end=" ")
    print("rep %d: %d blocks, layer fwd %.1f us; launch span %.0f ticks" % (rep, st.shape[0], 1e3 * ev0.elapsed_time(ev1), st[:, 7].max() - st[:, 0].min()))
    for k, nm in enumerate(names):
        if k != 5 and rep > 1:
            continue
        print("   %-14s mean %8.0f  p50 %8.0f  p95 %8.0f  max %8.0f ticks" % (nm, d[:, k].mean(), np.percentile(d[:, k], 50), np.percentile(d[:, k], 95), d[:, k].max()))
    tot = (st[:, 7] - st[:, 0]).astype(np.float64)
    print("   block total    mean %8.0f  max %8.0f;  start offsets p50 %.0f max %.0f" % (tot.mean(), tot.max(), np.percentile(st[:, 0] - st[:, 0].min(), 50), (st[:, 0] - st[:, 0].min()).max()))
