"""Timing probe for temp_segment_sum_rows on the loss path's shapes (development tool)."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from temp_amd import backend as TB, functional as TF
dev = torch.device("cuda:0")
be = TB.get_backend()
rng = np.random.default_rng(0)
def bench(name, n_seg, n_rows, d, idx):
    inv = TF.gather_inverse(idx, n_seg, dev)
    src = torch.randn(n_rows, d, device=dev)
    for _ in range(3): be.segment_sum_rows(src, inv[0], inv[1], n_seg)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): out = be.segment_sum_rows(src, inv[0], inv[1], n_seg)
    b.record(); torch.cuda.synchronize()
    ref = torch.zeros(n_seg, d, device=dev, dtype=torch.float64).index_add_(0, torch.from_numpy(idx).to(dev), src.double())
    err = float((out.double() - ref).abs().max())
    print("%-34s n_seg %6d rows %7d: %7.1f us   (%.0f GB/s)  max err %.2e" % (name, n_seg, n_rows, a.elapsed_time(b) / 20 * 1e3, n_rows * d * 4 / (a.elapsed_time(b) / 20 * 1e-3) / 1e9, err))
bench("rel rows (20 of 40 used)", 40, 48000, 200, rng.integers(0, 20, 48000))
bench("known rows", 4000, 48000, 200, rng.integers(0, 4000, 48000))
bench("embedding table (hot)", 500, 81000, 200, rng.integers(0, 500, 81000))
bench("visits -> distinct", 81000, 116000, 200, rng.integers(0, 81000, 116000))
bench("one segment", 1, 48000, 200, np.zeros(48000, np.int64))
