"""Verdict r5 item 6: is k_gru_wgrad power-limited?  Samples socket power and shader clock (sysfs hwmon, >= 20 Hz; rocm-smi as a
fallback) while the weight-gradient kernel runs back to back for a few seconds -- f16 (three products) and bf16 (six products)
arithmetic -- and while the chip idles.  Prints one table; run on the GPU box, commit the output under profiles/."""
import glob
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
from temp_amd import _lib
from temp_amd import backend as TB


def read(path, scale):
    if path is None:
        return None
    try:
        return float(open(path).read().strip()) * scale
    except Exception:
        return None


_PICK = {}


def sensors():
    """hwmon files of OUR card: sysfs lists every card of the node (other tenants' too); ours is the one at device 0's PCI address."""
    if _PICK:
        return _PICK["pw"], _PICK["fq"], _PICK["cap"]
    import ctypes
    import os
    cards = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
    buf = ctypes.create_string_buffer(64)
    hip = ctypes.CDLL("libamdhip64.so")
    rc = hip.hipDeviceGetPCIBusId(buf, 64, 0)
    bus = buf.value.decode().lower() if rc == 0 else ""
    c = None
    for cand in cards:
        pci = os.path.basename(os.path.realpath(cand.split("/hwmon/")[0])).lower()
        if bus and pci.endswith(bus[-12:]):
            c = cand
    print("device 0 is PCI %s -> %s (of %d cards in sysfs)" % (bus, c, len(cards)))
    if c is None and cards:
        c = cards[0]
    pw = None
    if c:
        for name in ("power1_average", "power1_input"):
            if glob.glob(c + "/" + name):
                pw = c + "/" + name
                break
    _PICK.update(pw=pw, fq=(c + "/freq1_input") if c else None, cap=(c + "/power1_cap") if c else None)
    return _PICK["pw"], _PICK["fq"], _PICK["cap"]


def smi():
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
        p = [l for l in out.splitlines() if "Power" in l and "W" in l]
        c = [l for l in out.splitlines() if "sclk" in l]
        return (p[0].strip() if p else ""), (c[0].strip() if c else "")
    except Exception as e:
        return str(e), ""


def sample_while(fn, seconds):
    pw, fq, cap = sensors()
    rows, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            rows.append((time.perf_counter(), read(pw, 1e-6) if pw else None, read(fq, 1e-6) if fq else None))
            time.sleep(0.04)
    th = threading.Thread(target=poll)
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        n += 50
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    tail = rows[len(rows) // 3:]                      # steady state: the last two thirds
    p = [r[1] for r in tail if r[1] is not None]
    f = [r[2] for r in tail if r[2] is not None]
    return dict(launches=n, us_per_launch=1e6 * dt / max(n, 1), samples=len(rows), hz=len(rows) / dt,
                power_w=(min(p), sum(p) / len(p), max(p)) if p else None, sclk_mhz=(min(f), sum(f) / len(f), max(f)) if f else None,
                smi=smi() if not p else None)


def main():
    dev = torch.device("cuda:0")
    be = TB.get_backend()
    lib = _lib.load()
    d, rows = 200, (60000, 58000)
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(n, d, generator=g).to(dev) for n in rows]
    hd = [(torch.rand(n, d, generator=g) * 2 - 1).to(dev) for n in rows]
    g4 = [(torch.randn(n, 4 * d, generator=g) * 0.1).to(dev) for n in rows]
    ws = [((torch.rand(3 * d, d, generator=g) - 0.5) * 0.3).to(dev) for _ in rows]
    rk = [t[:, :3 * d].abs().max(dim=1).values.view(torch.int32).contiguous() for t in g4]
    ck = [t.abs().max(dim=0).values.view(torch.int32).contiguous() for t in g4]
    xk = [be.absmax_keys(t, rows=False)[1].contiguous() for t in xs]
    none = [None, None]
    pw, fq, cap = sensors()
    print("sensors: power %s  clock %s  cap %s W" % (pw, fq, read(cap, 1e-6) if cap else None))
    print("idle:", sample_while(lambda: None, 2.0))
    print("k_gru_wgrad_hx (f16, three products) back to back:", sample_while(lambda: be.gru_grads_g4(xs, hd, g4, ws, none, row_keys=rk, col_keys=ck, x_col_keys=xk), 6.0))
    lib.temp_set_option(_lib.OPT_MFMA_F16X2, 0)
    print("k_gru_wgrad (bf16, six products) back to back:", sample_while(lambda: be.gru_grads_g4(xs, hd, g4, ws, none), 6.0))
    lib.temp_set_option(_lib.OPT_MFMA_F16X2, 1)
    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    print("torch bf16 8192^3 matmul back to back (a power-hungry reference):", sample_while(lambda: a @ a, 4.0))


if __name__ == "__main__":
    main()
