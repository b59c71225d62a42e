#!/usr/bin/env python3
"""MFMA utilisation per kernel from a rocprofv3 --pmc counter_collection.csv holding SQ_VALU_MFMA_BUSY_CYCLES and
GRBM_GUI_ACTIVE (+ optional SQ_INSTS_VALU_MFMA_MOPS_BF16 / _F32 / _F16).  GRBM_GUI_ACTIVE is summed over the 8 XCDs and
SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs: mfma_util = busy / ((gui / 8) * 1024); flops = MOPS * 512.
usage: pmc_mfma.py <counter_collection.csv> [name-substring]"""
import collections, csv, json, sys
path = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
tot = collections.defaultdict(lambda: collections.Counter())
cnt = collections.defaultdict(lambda: collections.Counter())
with open(path) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        if pat and pat not in n:
            continue
        tot[n][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[n][r["Counter_Name"]] += 1
out = {}
for n, c in tot.items():
    if "GRBM_GUI_ACTIVE" not in c or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
        continue
    d = cnt[n]["GRBM_GUI_ACTIVE"]
    gui = c["GRBM_GUI_ACTIVE"] / d / 8.0
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / cnt[n]["SQ_VALU_MFMA_BUSY_CYCLES"]
    mops = sum(c[k] / cnt[n][k] for k in c if k.startswith("SQ_INSTS_VALU_MFMA_MOPS"))
    rec = dict(dispatches=d, mean_cycles=round(gui), mfma_util=round(busy / (gui * 1024.0), 3) if gui else None)
    if mops:
        rec["mfma_flops_per_dispatch"] = mops * 512
    out[n[:90]] = rec
print(json.dumps(out, indent=1))
