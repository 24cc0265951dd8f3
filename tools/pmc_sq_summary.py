#!/usr/bin/env python3
"""Per-kernel SQ counters of one rocprofv3 --pmc pass (counters only, eager launches) as ratios:

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES \
              SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d <dir> -o pmc -- python bench.py ...
    python tools/pmc_sq_summary.py <dir>/.../pmc_counter_collection.csv [profiles/<tag>_bench_kernel_stats.csv] > profiles/<tag>_pmc_sq.md

MI355X_MICROARCH.md (rocprofv3 PMC slots): WAIT_ANY (wave parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY
~ WAVE_CYCLES (quad-cycles, disjoint); WAIT_INST_LDS is a sub-bucket of WAIT_INST_ANY; VALU_MFMA_BUSY_CYCLES counts cycles of the matrix
pipe (32 per 32x32x16 MFMA) summed over the SIMDs.  `mfma pipe` = MFMA busy cycles per launch / (the kernel's average duration in the
rocprofv3 kernel statistics given as second argument x 2.0 GHz x 1024 SIMDs): the share of the chip's matrix-pipe cycles the kernel
fills at the clock it typically gets (GRBM_GUI_ACTIVE is summed over the XCDs' instances and is printed raw); LDS conflict =
BANK_CONFLICT / IDX_ACTIVE."""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("temp::", "")


rows = list(csv.DictReader(open(sys.argv[1])))
dur_ns = {}
if len(sys.argv) > 2:
    for r in csv.DictReader(open(sys.argv[2])):
        dur_ns[short(r["Name"])] = float(r["AverageNs"])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
seen = set()
for r in rows:
    k = short(r["Kernel_Name"])
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r.get("Dispatch_Id"), k)
    if key not in seen:
        seen.add(key)
        calls[k] += 1
names = sorted(acc, key=lambda k: -acc[k].get("GRBM_GUI_ACTIVE", acc[k].get("SQ_WAVE_CYCLES", 0.0)))
print("| kernel | launches | avg us (kernel stats) | parked (WAIT_ANY) | issue stall (WAIT_INST_ANY) | of it LDS | issuing (ACTIVE_INST_ANY) | MFMA busy cycles / launch | mfma pipe | LDS bank conflict |")
print("|---|---|---|---|---|---|---|---|---|---|")
for k in names[:24]:
    c = acc[k]
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    f = lambda n: "%.0f %%" % (100.0 * c.get(n, 0.0) / wc)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(calls[k], 1)
    d = dur_ns.get(k)
    mf = "%.0f %%" % (100.0 * busy / (d * 2.0 * 1024.0)) if (d and busy > 0) else "-"
    lds = "%.0f %%" % (100.0 * c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]) if c.get("SQ_LDS_IDX_ACTIVE", 0.0) > 0 else "-"
    print("| `%s` | %d | %s | %s | %s | %s | %s | %.3g | %s | %s |" % (k[:60], calls[k], ("%.1f" % (d / 1e3)) if d else "-", f("SQ_WAIT_ANY"), f("SQ_WAIT_INST_ANY"),
                                                                   f("SQ_WAIT_INST_LDS"), f("SQ_ACTIVE_INST_ANY"), busy, mf, lds))
