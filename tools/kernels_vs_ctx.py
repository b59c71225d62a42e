#!/usr/bin/env python3
"""Duration of every kernel class of the decode step against the context length, from a rocprofv3 kernel trace of one generation
(see tools/attn_vs_ctx.py): does a kernel slow down as the attention before it streams more KV (cache / TLB state), or not?"""
import csv, sys, statistics, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
attn = [i for i, r in enumerate(rows) if "attn_paged_kernel" in r["Kernel_Name"]]
NL, n_steps = 6, 1019
first = attn[-NL * n_steps]                       # first attention launch of the last generation
seq = rows[first - 1:]                            # starts at that step's QKV kernel
per_step = 32
buckets = collections.defaultdict(lambda: collections.defaultdict(list))
names = {}
for i, r in enumerate(seq[: per_step * n_steps]):
    s, k = divmod(i, per_step)
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    kind = ("L%d:" % (k // 5) if k < 30 else "") + r["Kernel_Name"].split("(")[0].replace("void mgea::", "")[:44]
    pos = k % 5 if k < 30 else 5 + (k - 30)
    names[pos] = r["Kernel_Name"].split("(")[0].replace("void mgea::", "")[:48]
    buckets[pos][(5 + s + 1) // 128].append(d)
print("position in layer -> median us by context bucket of 128 tokens")
for pos in sorted(buckets):
    print(f"{pos} {names[pos]:48s} " + " ".join(f"{statistics.median(buckets[pos][b]):6.2f}" for b in sorted(buckets[pos])))

# per (layer, position) medians and spread over the whole generation
print("\nper layer: median / p10 / p90 us")
per = collections.defaultdict(list)
for i, r in enumerate(seq[: per_step * n_steps]):
    s, k = divmod(i, per_step)
    per[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
gaps = collections.defaultdict(list)
for i in range(1, per_step * n_steps):
    gaps[i % per_step].append((int(seq[i]["Start_Timestamp"]) - int(seq[i - 1]["End_Timestamp"])) / 1e3)
for k in range(per_step):
    v = sorted(per[k]); g = sorted(gaps[k])
    nm = seq[k]["Kernel_Name"].split("(")[0].replace("void mgea::", "")[:40]
    print(f"{k:2d} {nm:40s} {v[len(v)//2]:6.2f} {v[len(v)//10]:6.2f} {v[9*len(v)//10]:6.2f}   gap before: {g[len(g)//2]:5.2f}")
