#!/usr/bin/env python3
"""Launches of a rocprofv3 kernel trace grouped by (kernel, blocks, block size): total / average duration -- the under-filled
launches (few blocks, tens of microseconds) are latency chains on an idle chip.   python tools/trace_underfilled.py <kernel_trace.csv> [steps]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
d = collections.defaultdict(list)
for r in rows:
    wg = int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z'])
    g = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
    d[(r['Kernel_Name'][:78], g // wg, wg)].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
out = sorted(((sum(v) / 1e3, k, len(v)) for k, v in d.items()), reverse=True)
for tot, k, n in out[:int(sys.argv[3]) if len(sys.argv) > 3 else 60]:
    print("%9.1f us/step  %-80s blocks %6d x %4d thr  n/step %5.1f  avg %7.1f us" % (tot / steps, k[0], k[1], k[2], n / steps, tot / n))
