#!/usr/bin/env python3
"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs,
counters only -- MI355X_MICROARCH.md, HBM section).

    python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE/pmc_counter_collection.csv \
                                gpurun_out/pmc_WRITE_SIZE/pmc_counter_collection.csv profiles/r01_pmc_traffic.json

Units / corrections (same guide): both counters are reported in KiB; on gfx950 FETCH_SIZE counts 64 B per
128-B request for wide (16 B / lane) coalesced reads, so it is DOUBLED here; WRITE_SIZE is uncalibrated and
kept raw.  Infinity-Cache hits are counted (the counters sit on the L2's fabric side)."""
import collections
import csv
import json
import re
import sys


# rocprof kernel names -> the family names of bench.py's event trace (one trace id per call site, whichever kernel serves it)
FAMILY_ALIAS = {"k_gemm_tn_bx": "k_gemm_tn", "k_gemm_tn_bx8": "k_gemm_tn", "k_rgcn_agg_s": "k_rgcn_agg", "k_rgcn_dw_s": "k_rgcn_dw", "k_gemm_bxp": "k_gemm_panel", "k_gemm_bx": "k_gemm_panel", "k_gemm_wres": "k_gemm_panel", "k_gemm_bxr": "k_gemm_panel", "k_rgcn_agg_t": "k_rgcn_agg"}


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("temp::", "")


def agg(path):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        d[k][0] += 1
        d[k][1] += float(r["Counter_Value"])
    return d


def main():
    f, w = agg(sys.argv[1]), agg(sys.argv[2])
    out = {}
    for k in f:
        if not k.startswith("k_"):
            continue
        calls = f[k][0]
        fetch = 2.0 * 1024.0 * f[k][1] / calls
        wk = w.get(k, [1, 0.0])
        write = 1024.0 * wk[1] / max(wk[0], 1)
        out[k] = dict(launches=calls, fetch_bytes_per_launch=fetch, write_bytes_per_launch_raw=write,
                      traffic_bytes_per_launch=fetch + write)
    fam = collections.defaultdict(lambda: [0, 0.0])           # bench.py's kernel-family names (template args dropped)
    for k, v in out.items():
        base = re.sub(r"<.*$", "", k)
        base = FAMILY_ALIAS.get(base, base)
        fam[base][0] += v["launches"]
        fam[base][1] += v["traffic_bytes_per_launch"] * v["launches"]
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else None        # encoder steps the profiled process executed
    chain = sum(v["launches"] for k, v in out.items() if k.startswith("k_gru_chain_fwd"))
    if chain:                                                      # one forward chain launch per encoder step: count them instead of trusting the caller
        steps = chain
    doc = dict(source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 3 --warmup 1 --no-graph",
               corrections="FETCH_SIZE x2 (gfx950 wide-read under-count), WRITE_SIZE raw; KiB -> bytes",
               kernels=out, families={k: dict(launches=v[0], traffic_bytes_per_launch=v[1] / v[0]) for k, v in fam.items()})
    if steps:
        doc["steps_in_run"] = steps
        doc["step_traffic_bytes"] = sum(v["traffic_bytes_per_launch"] * v["launches"] for v in out.values()) / steps
        doc["step_fetch_bytes"] = sum(v["fetch_bytes_per_launch"] * v["launches"] for v in out.values()) / steps
        doc["step_write_bytes_raw"] = sum(v["write_bytes_per_launch_raw"] * v["launches"] for v in out.values()) / steps
    json.dump(doc, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"]):
        print("%-44s launches %4d  fetch %9.1f MB  write(raw) %9.1f MB per launch" % (k[:44], v["launches"], v["fetch_bytes_per_launch"] / 1e6,
                                                                                       v["write_bytes_per_launch_raw"] / 1e6))


if __name__ == "__main__":
    main()
