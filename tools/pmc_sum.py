#!/usr/bin/env python3
"""Mean per-dispatch value of every counter in a rocprofv3 --pmc counter_collection.csv, by kernel:  pmc_sum.py <csv> [name-substring]"""
import collections, csv, json, sys
tot, cnt = collections.defaultdict(collections.Counter), collections.defaultdict(collections.Counter)
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if pat and pat not in n: continue
    tot[n][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[n][r["Counter_Name"]] += 1
print(json.dumps({n[:80]: {k: round(v / cnt[n][k], 1) for k, v in c.items()} | {"dispatches": max(cnt[n].values())} for n, c in tot.items()}, indent=1))
