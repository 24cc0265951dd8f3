#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls, total, average, share.
    python tools/rocpd_stats.py gpurun_out/prof/bench_results.db [--md] [--grid]
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("temp::", "")


def main():
    path = sys.argv[1]
    md = "--md" in sys.argv
    by_grid = "--grid" in sys.argv
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = cur.execute("select s.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x, d.end - d.start "
                       "from %s d join %s s on d.kernel_id = s.id" % (kd, ks)).fetchall()
    agg = {}
    for name, gx, gy, gz, wx, dur in rows:
        key = short(name) + ((" grid=%dx%dx%d/%d" % (gx // max(wx, 1), gy, gz, wx)) if by_grid else "")
        a = agg.setdefault(key, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values())
    hdr = ("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|" if md else
           "%-90s %7s %10s %9s %9s %9s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "%"))
    print(hdr)
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        vals = (k, a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / total)
        print(("| %s | %d | %.3f | %.1f | %.1f | %.1f | %.1f |" if md else "%-90s %7d %10.3f %9.1f %9.1f %9.1f %6.1f") % vals)
    print(("\ntotal kernel time: %.3f ms over %d dispatches" % (total / 1e6, len(rows))))


if __name__ == "__main__":
    main()
