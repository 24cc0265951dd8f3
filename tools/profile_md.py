#!/usr/bin/env python3
"""Write profiles/<tag>_bench_kernel_stats.{md,csv} from a rocprofv3 --kernel-trace --stats run of bench.py.
    python tools/profile_md.py <stats.csv> <bench_line.json> <tag> "<command>" <process_steps | auto>
"""
import csv
import json
import re
import shutil
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("temp::", "")


def main():
    stats, line, tag, cmd, steps = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5]
    d = json.loads(open(line).read().strip().splitlines()[-1])
    rows = list(csv.DictReader(open(stats)))
    # "auto": one k_gru_chain_fwd launch per encoder step of the process
    steps = sum(int(x["Calls"]) for x in rows if "k_gru_chain_fwd" in x["Name"]) if steps == "auto" else int(steps)
    shutil.copy(stats, "profiles/%s_bench_kernel_stats.csv" % tag)
    json.dump(d, open("profiles/%s_bench_line.json" % tag, "w"), indent=1)
    r = d["roofline"]
    cpu = d.get("cpu_baseline") or {}
    out = ["# %s — rocprofv3 kernel-trace summary of the default bench" % tag, "",
           "Command (on the MI355X box): `%s`  " % cmd,
           "Raw stats: `profiles/%s_bench_kernel_stats.csv`; bench JSON line of the same run: `profiles/%s_bench_line.json`." % (tag, tag), "",
           "Bench line: **%.1f M edge visits/s, %.2f ms/step** (%s %s L=%d bsz=%d D=%d, %s); `roofline.kernel` = `%s` at %.1f %s "
           "(%.0f%% of the %.1f peak), bench HIP-event average %.1f us per launch." % (
               d["value"] / 1e6, d["ms_per_step"], d["config"]["workload"], d["config"]["encoder"], d["config"]["seq_len"],
               d["config"]["windows_per_gpu"], d["config"]["embed"], d["config"]["launch"], r["kernel"], r["achieved"], r["unit"],
               100 * r["frac"], r["peak"], 1e3 * r["avg_launch_ms"]), ""]
    if cpu:
        out += ["CPU baseline of the same run (oracle port, %d threads): %.0f edge visits/s  ->  GPU/CPU = %.0fx." % (
            cpu["cores"], cpu["value"], d["value"] / cpu["value"]), ""]
    out += ["The profiled process executes %d encoder steps in total (pre-capture + warm-up + timed graph replays + event-traced eager "
            "steps); 'ms/step' below = total / %d." % (steps, steps), "",
            "| kernel | calls | total ms | avg us | % | ms/step |", "|---|---|---|---|---|---|"]
    for x in rows:
        tot = float(x["TotalDurationNs"]) / 1e6
        if float(x["Percentage"]) < 0.3:
            continue
        out.append("| `%s` | %s | %.2f | %.1f | %s | %.3f |" % (short(x["Name"]), x["Calls"], tot, float(x["AverageNs"]) / 1e3, x["Percentage"], tot / steps))
    open("profiles/%s_bench_kernel_stats.md" % tag, "w").write("\n".join(out) + "\n")
    print("\n".join(out[:14]))


if __name__ == "__main__":
    main()
