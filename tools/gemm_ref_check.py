"""Calibration only: what does the vendor fp32 GEMM (torch.mm -> rocBLAS/hipBLASLt) reach on the
skinny shapes of this workload?  (A ceiling reference for the hand-written MFMA kernels.)"""
import torch, time
dev = torch.device("cuda")
def bench(fn, it=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
M = 116000
for (m, k, n, name) in [(M, 200, 200, "loop fwd  X.W"), (M, 200, 600, "gi   X.Wih^T"), (M, 600, 200, "dx   dgi.Wih"), (4000, 200, 600, "cell h.Whh^T"), (4000, 600, 200, "cell dprev")]:
    a = torch.randn(m, k, device=dev); b = torch.randn(k, n, device=dev)
    ms = bench(lambda: torch.mm(a, b))
    print("%-16s M=%6d K=%3d N=%3d  %.4f ms  %.1f TF/s" % (name, m, k, n, ms, 2.0 * m * k * n / ms / 1e9))
for (m, ka, nb, name) in [(M, 200, 200, "dWloop X^T.dZ"), (58000, 600, 200, "dWih dgi^T.x")]:
    a = torch.randn(m, ka, device=dev); b = torch.randn(m, nb, device=dev)
    ms = bench(lambda: torch.mm(a.t(), b))
    print("%-16s M=%6d Ka=%3d Nb=%3d  %.4f ms  %.1f TF/s" % (name, m, ka, nb, ms, 2.0 * m * ka * nb / ms / 1e9))
