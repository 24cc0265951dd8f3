#!/usr/bin/env python3
"""One step's kernel sequence from a rocprofv3 kernel trace: start offset, duration and the gap to the previous kernel, in
launch order.  The step is found as the shortest period of the kernel-name sequence at the end of the trace.
    python tools/step_sequence.py <kernel_trace.csv> [min_period]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 20
P = None
for p in range(lo, len(names) // 2):
    if names[-p:] == names[-2 * p:-p]:
        P = p
        break
if P is None:
    print("no period found; last 120 kernels")
    P = min(120, len(names))
seq = rows[-P:]
t0 = int(seq[0]['Start_Timestamp'])
prev_end = t0
tot = gaps = 0.0
for r in seq:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    wg = int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z'])
    g = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
    print("%9.1f  dur %7.1f  gap %6.1f  blocks %6d x %4d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, g // wg, wg, r['Kernel_Name'][:100]))
    tot += (e - s) / 1e3
    gaps += max(0, s - prev_end) / 1e3
    prev_end = max(prev_end, e)
print("period %d kernels: kernel time %.1f us, gaps %.1f us, span %.1f us" % (P, tot, gaps, (prev_end - t0) / 1e3))
