#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv: per-kernel mean of a counter per dispatch.
usage: pmc_traffic.py <counter_collection.csv> <COUNTER> [name-substring]"""
import csv, os, subprocess, sys, collections, json
path, counter = sys.argv[1], sys.argv[2]
pat = sys.argv[3] if len(sys.argv) > 3 else ""
tot, cnt = collections.Counter(), collections.Counter()
with open(path) as f:
    for r in csv.DictReader(f):
        if r.get("Counter_Name") != counter:
            continue
        name = r["Kernel_Name"]
        if pat and pat not in name:
            continue
        tot[name] += float(r["Counter_Value"]); cnt[name] += 1
out = {k[:80]: dict(dispatches=cnt[k], n=cnt[k], mean=tot[k] / cnt[k], total=tot[k]) for k in tot}
# provenance for bench.py's roofline.traffic_source: the commit the pass ran on and the command that produced the CSV
try:
    out["_commit"] = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL, text=True).strip()
except Exception:
    out["_commit"] = os.environ.get("MGEA_COMMIT", "unknown")     # the GPU box has no .git: profile_round.sh passes it in
out["_command"] = os.environ.get("MGEA_PMC_COMMAND", "unknown")
out["_counter"] = counter
print(json.dumps(out, indent=1))
