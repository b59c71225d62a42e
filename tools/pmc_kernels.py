#!/usr/bin/env python3
"""Per-kernel means of every counter in a rocprofv3 --pmc counter_collection.csv, keyed by (kernel, grid size).
usage: pmc_kernels.py <counter_collection.csv> [name-substring]"""
import csv, sys, collections
path = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
tot, cnt = collections.defaultdict(float), collections.Counter()
with open(path) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if pat and pat not in name: continue
        key = (name[:60], r.get("Grid_Size", ""), r["Counter_Name"])
        tot[key] += float(r["Counter_Value"]); cnt[key] += 1
for k in sorted(tot):
    print(f"{k[0]:60s} grid {k[1]:>8s} {k[2]:28s} mean {tot[k] / cnt[k]:16.1f}  (n = {cnt[k]})")
