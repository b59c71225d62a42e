#!/usr/bin/env python3
"""From a rocprofv3 kernel_trace.csv: gaps between consecutive kernels, split into gaps inside a decode step's graph and
gaps across graph launches (after the argmax/embed tail kernel).  usage: trace_gaps.py <kernel_trace.csv>"""
import csv, sys, statistics
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
inner, outer = [], []
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gap = (s1 - e0) / 1e3
    if gap < 0 or gap > 200:
        continue
    (outer if "argmax_advance_embed" in n0 else inner).append(gap)
for name, g in (("inside a step", inner), ("between steps (after the tail kernel)", outer)):
    if g:
        print(f"{name:40s}: n={len(g):6d} median {statistics.median(g):6.2f} us  mean {statistics.mean(g):6.2f} us  p90 {sorted(g)[int(0.9*len(g))]:6.2f} us")
