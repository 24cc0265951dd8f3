#!/usr/bin/env python3
"""Development probe: the torch operators (not library kernels) inside ONE eager training step of a bench configuration -- the
additions autograd inserts where a tensor has several consumers, zero fills of slice / select backward, concatenations, copies --
with their shapes and counts.  What a fused node or a whole-tensor hand-over can remove shows up here.
    python tools/glue_audit.py [headline|with_loss|attention|config1_static|config2_default_flags|config3_post_ensemble]"""
import collections
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from temp_amd import synthetic  # noqa: E402
from temp_amd.sampling import CorruptTriples  # noqa: E402


def build(name, device):
    if name in ("headline", "with_loss", "attention"):
        w = synthetic.workload("S-gdelt", seed=0)
        m = bench.build_model(w, device, "attention" if name == "attention" else "gru")
        m.sample_rng = np.random.default_rng(2)
        wb = m.prepare(synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 0), w["L"], train=True)
        if name == "with_loss":
            m.corrupter = CorruptTriples(m.args, w["snapshots"], seed=5)
            fixed = [tuple(x.to(device) for x in smp) for smp in m.draw_samples(wb)]
            return m, lambda: m.run_loss(wb, fixed)
        return m, lambda: m.run(wb)[0].sum()
    w = synthetic.workload("S-icews0515" if name == "config3_post_ensemble" else "S-icews14", seed=0)
    if name == "config1_static":
        from temp_amd.static_rgcn import StaticRGCN
        m = StaticRGCN(bench.make_args(w, "SRGCN"), w["num_ents"], w["num_rels"], w["snapshots"], w["snapshots"], w["snapshots"]).to(device)
    elif name == "config2_default_flags":
        from temp_amd.dynamic_rgcn import DynamicRGCN
        args = bench.make_args(w, "GRRGCN")
        args.rec_only_last_layer = False
        m = DynamicRGCN(args, w["num_ents"], w["num_rels"], w["snapshots"], w["snapshots"], w["snapshots"]).to(device)
    else:
        from temp_amd.post_dynamic_rgcn import PostEnsembleBiDynamicRGCN
        args = bench.make_args(w, "BiGRRGCN")
        args.post_ensemble = True
        m = PostEnsembleBiDynamicRGCN(args, w["num_ents"], w["num_rels"], w["snapshots"], w["snapshots"], w["snapshots"]).to(device)
    m.sample_rng = np.random.default_rng(2)
    m.corrupter = CorruptTriples(m.args, w["snapshots"], seed=5)
    targets = synthetic.default_targets(w["num_times"], w["L"], w["bsz"], 3)
    if name == "config1_static":
        wb = m.prepare(targets)
        with torch.no_grad():
            m.run_loss(wb)
        cand = m._last_plan[1]
        return m, lambda: m.run_loss(wb, cand)
    wb = m.prepare(targets, w["L"], True)
    fixed = [tuple(x.to(device) for x in smp) for smp in m.draw_samples(wb)]
    if name == "config2_default_flags":
        return m, lambda: m.run_loss(wb, fixed)
    wts = [(torch.full((smp[0].shape[0], 1), 0.5, device=device), torch.full((smp[0].shape[0], 1), 0.5, device=device)) for smp in fixed]
    return m, lambda: m.run_loss(wb, fixed, wts)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "headline"
    device = torch.device("cuda", 0)
    torch.manual_seed(1)
    m, fn = build(name, device)

    def step():
        for p in m.parameters():
            p.grad = None
        fn().backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        t = getattr(e, "self_device_time_total", None)
        if t is None:
            t = getattr(e, "self_cuda_time_total", 0.0)
        if e.key.startswith("aten::") and t > 0:
            rows.append((t, e.count, e.key, str(e.input_shapes)[:90]))
    total = sum(r[0] for r in rows)
    print("%s: %d torch operator calls with device time of their own, %.1f us on the device in all" % (name, sum(r[1] for r in rows), total))
    for t, n, k, shp in sorted(rows, key=lambda r: -r[0])[:25]:
        print("  %7.1f us  x%-3d %-18s %s" % (t, n, k, shp))


if __name__ == "__main__":
    main()
