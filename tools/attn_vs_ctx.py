#!/usr/bin/env python3
"""Paged-attention launch duration against the context length, from a rocprofv3 kernel trace of one generation
(rocprofv3 --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-extra --profile-stride 0):
the launches of the generation are consecutive, 6 per decode step; step s has ctx = prompt + s + 1."""
import csv, sys, statistics
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "attn_paged_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1019
rows = rows[-NL * n_steps:]                      # the last generation
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("ctx   pages  us/launch (median over the layers and the 64-token window)")
for p in range(0, 16):
    lo, hi = p * 64, p * 64 + 64
    for q in range(4):                              # quarters of the page range
        a, b = lo + q * 16, lo + q * 16 + 16
        v = [dur[s * NL + l] for s in range(n_steps) for l in range(NL) if a <= 5 + s + 1 < b]
        if v:
            print(f"{a:4d}..{b:4d}  {p + 1:2d}  {statistics.median(v):7.2f}")
