#!/usr/bin/env python3
"""One step's kernel sequence from a rocprofv3 kernel trace of bench.py: start offset, duration and the gap to the previous
kernel's end, in launch order.  A step is delimited by a marker kernel that runs once per step (default k_gru_chain_fwd); the
step printed is one from the MIDDLE of the trace (the HIP-graph replays of the timed region), rotated to start behind the largest
gap of the period (the gap between two graph launches).  Kernels of parallel graph branches may swap places from step to step, so
the sequence is not required to repeat name by name.
    python tools/step_sequence.py <kernel_trace.csv> [marker substring]"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
marker = sys.argv[2] if len(sys.argv) > 2 else "k_gru_chain_fwd"
m = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
if len(m) < 4:
    sys.exit("marker kernel %r runs fewer than four times in this trace" % marker)
k = len(m) // 2
a, b = m[k], m[k + 1]                     # one period, marker to marker
P = b - a
# rotate: the step starts behind the largest gap inside [a - P, a]
best, start = -1, a
for i in range(a - P + 1, a + 1):
    gap = int(rows[i]['Start_Timestamp']) - max(int(r['End_Timestamp']) for r in rows[max(0, i - 4):i])
    if gap > best:
        best, start = gap, i
seq = rows[start:start + P]
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
print("period %d kernels (gap before it %.1f us): kernel time %.1f us, gaps %.1f us, span %.1f us" % (P, best / 1e3, tot, gaps, (prev_end - t0) / 1e3))
