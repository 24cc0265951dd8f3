#!/usr/bin/env python3
"""Development probe: the segment-sum kernels on SKEWED segmentations (the loss path's known-entity gather adjoint: 48 000 rows
into 8 x 500 table rows with Zipf hubs).  The n_rows hint selects the kernel: real (wave per segment), 33 x n_seg (block of 4
waves per segment), 97 x n_seg (16 waves)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from temp_amd import _lib, functional as TF
from temp_amd.backend import get_backend, _ptr
lib = _lib.load()
be = get_backend()
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for n_seg, n_rows, a in ((4000, 48000, 1.2), (4000, 48000, 0.0), (60000, 120000, 0.0), (7691, 48000, 1.1)):
    if a > 0:
        p = 1.0 / np.arange(1, n_seg + 1) ** a
        p /= p.sum()
        idx = rng.choice(n_seg, size=n_rows, p=p)
    else:
        idx = rng.integers(0, n_seg, n_rows)
    seg_ptr, order = TF.gather_inverse(idx, n_seg, dev)
    src = torch.randn(n_rows, 200, device=dev)
    out = torch.empty(n_seg, 200, device=dev)
    ref = torch.zeros(n_seg, 200, device=dev, dtype=torch.float64).index_add_(0, torch.from_numpy(idx).to(dev), src.double())
    print("n_seg %d rows %d zipf %.1f longest segment %d" % (n_seg, n_rows, a, int(np.bincount(idx, minlength=n_seg).max())))
    for label, hint in (("auto", n_rows), ("blk4", 33 * n_seg), ("blk16", 97 * n_seg)):
        nb = lib.temp_segment_sum_rows_workspace(n_seg, hint, 200)
        ws = torch.empty(max(nb, 4) // 4, device=dev) if nb else None
        f = lambda: lib.temp_segment_sum_rows(n_seg, hint, 200, _ptr(seg_ptr), _ptr(order), _ptr(src), _ptr(out), _ptr(ws), nb, None)
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): f()
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 20
        print("   %-6s %7.1f us  max err %.1e" % (label, 1e6 * t, float((out.double() - ref).abs().max())))
