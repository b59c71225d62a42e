#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of tools/bert_prof.py: the bf16 DistilBERT forward kernel by kernel -- median duration of every
launch position of the LAST forwards (the layer structure repeats: QKV, attention, out-proj, row statistics, FC1, FC2, row statistics).
usage: bert_layer_times.py <kernel_trace.csv> [launches per forward, default: detected from the embed kernel]"""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "bert_embed_ln_bf16_kernel" in n]
assert len(starts) >= 3, "need a few forwards in the trace"
fw = [rows[a:b] for a, b in zip(starts[-4:-1], starts[-3:])]          # three complete forwards
L = min(len(f) for f in fw)
tot = []
print("pos  kernel                                              median us   (three forwards)")
for k in range(L):
    d = [(int(f[k]["End_Timestamp"]) - int(f[k]["Start_Timestamp"])) / 1e3 for f in fw]
    n = fw[0][k]["Kernel_Name"].replace("void ", "").replace("mgea::", "")[:50]
    tot.append(statistics.median(d))
    print(f"{k:3d}  {n:50s} {statistics.median(d):9.2f}   {[round(x, 1) for x in d]}")
span = [(int(f[L - 1]["End_Timestamp"]) - int(f[0]["Start_Timestamp"])) / 1e3 for f in fw]
print(f"sum of kernel durations {sum(tot):.1f} us; forward span first start -> last end {statistics.median(span):.1f} us (gaps {statistics.median(span) - sum(tot):.1f} us)")
