import ctypes, sys, torch
sys.path.insert(0, ".")
from temp_amd import _lib, backend as TB
be = TB.get_backend(); lib = _lib.load()
dev = torch.device("cuda:0")
n, d = 82000, 200
g = torch.Generator().manual_seed(1)
e = torch.randn(n, d, generator=g).to(dev); w = (torch.randn(d, d, generator=g) * 0.1).to(dev); b = torch.randn(d, generator=g).to(dev)
def run():
    return be.rgcn_isolated_fwd(e, w, b, 1, None) if hasattr(be, "rgcn_isolated_fwd") else None
import inspect
print([m for m in dir(be) if "isolated" in m])
for it in range(3): out = run()
torch.cuda.synchronize()
lib.temp_trace_begin(512)
for it in range(10): out = run()
ids, ms, cnt = (ctypes.c_int32 * 512)(), (ctypes.c_float * 512)(), ctypes.c_int32(0)
lib.temp_trace_end(ids, ms, 512, ctypes.byref(cnt))
agg = {}
for i in range(cnt.value):
    k = lib.temp_trace_kernel_name(ids[i]).decode(); agg.setdefault(k, []).append(ms[i])
print({k: (len(v), round(1e3 * sum(v) / len(v), 1)) for k, v in agg.items()}, "f16 launches", lib.temp_f16_launches())
