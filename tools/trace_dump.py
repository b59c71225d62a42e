import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
# find an argmax kernel in the middle
idx = [i for i, r in enumerate(rows) if "argmax_advance_embed" in r[2]]
i0 = idx[len(idx) // 2]
for a, b in zip(rows[i0 - 3:i0 + 70], rows[i0 - 2:i0 + 71]):
    print(f"{a[2]:40s} dur {(a[1]-a[0])/1e3:6.2f}  gap-> {(b[0]-a[1])/1e3:6.2f}  q={a[3]} s={a[4]}")
