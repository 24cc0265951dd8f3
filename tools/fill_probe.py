#!/usr/bin/env python3
"""GPU box: which zero fills does a step of the HBM-regime window issue?  (verdict r4 item 3a: 12 ms of fillBufferAligned per
0.23-s step.)  torch.profiler with shapes over one eager step of the S-hbm-window workload at 2^k nodes: every aten::zero_ / fill_
with its tensor size and the Python frame that asked for it; then the library's own hipMemsetAsync sizes for that shape.
    python tools/fill_probe.py [--log2-nodes 14]"""
import argparse, os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench
from temp_amd import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--log2-nodes", type=int, default=14)
a = ap.parse_args()
k = a.log2_nodes
N, E, R, D, B, L = 1 << k, 1 << (k + 4), 230, 200, 100, 15
dev = torch.device("cuda:0")
snaps = synthetic.make_snapshots(N, R, E, N, 2 * L - 1, seed=0)
w = dict(name="S-hbm-window", num_ents=N, num_rels=R, edges_per_snap=E, nodes_per_snap=N, num_times=2 * L - 1, D=D, B=B, L=L, bsz=1, module="BiGRRGCN", snapshots=snaps)
model = bench.build_model(w, dev)
model.sample_rng = np.random.default_rng(2)
wb = model.prepare([L - 1], L, train=True)
st = bench.GraphStep(lambda: model.run(wb)[0], list(model.parameters()), graph=False)
for _ in range(2):
    st.eager()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
from temp_amd import _lib
lib = _lib.load()
OPT_OVERLAP = getattr(_lib, 'OPT_OVERLAP', None)
for overlap in (1, 0):
  lib.temp_set_option(OPT_OVERLAP, overlap)
  st.eager(); torch.cuda.synchronize()
  print("---- TEMP_OPT_OVERLAP = %d (d/dweight on the side stream beside d/dh: %s)" % (overlap, "yes" if overlap else "no"))
  with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
      st.eager()
      torch.cuda.synchronize()
  rows = []
  for ev in prof.events():
      if ev.name in ("aten::zero_", "aten::fill_", "aten::zeros", "aten::zeros_like", "aten::new_zeros", "aten::index_put_", "aten::copy_"):
          shp = ev.input_shapes[0] if ev.input_shapes else None
          n = int(np.prod(shp)) if shp else 0
          if ev.name in ("aten::zero_", "aten::fill_") and n * 4 >= 1 << 20:
              frames = [f for f in (ev.stack or []) if "temp_amd" in f or "bench.py" in f]
              rows.append((n * 4 / 2 ** 20, ev.name, shp, frames[:2], ev.device_time_total if hasattr(ev, "device_time_total") else ev.cuda_time_total))
  rows.sort(key=lambda r: -r[0])
  print("node rows N_total = %d  (one [N, D] fp32 tensor = %.1f MiB)" % (wb.n_node_visits, wb.n_node_visits * D * 4 / 2 ** 20))
  for mb, name, shp, frames, t in rows:
      print("%9.1f MiB  %-12s %-22s dev %8.1f us  %s" % (mb, name, shp, t, " <- ".join(f.strip() for f in frames)))
  ker = {}
  for ev in prof.events():
      if ev.device_type == torch.autograd.DeviceType.CUDA and ("fill" in ev.name.lower() or "memset" in ev.name.lower()):
          ker.setdefault(ev.name[:60], []).append(ev.device_time_total if hasattr(ev, "device_time_total") else ev.cuda_time_total)
  for kname, ts in ker.items():
      print("device: %-60s x%-3d total %.1f us  max %.1f us" % (kname, len(ts), sum(ts), max(ts)))
