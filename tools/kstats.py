#!/usr/bin/env python3
"""Print a rocprofv3 *_kernel_stats.csv compactly: name, calls, avg/min/max us, share."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 12]:
    if pat and pat not in r["Name"]:
        continue
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>7s} avg={float(r['AverageNs'])/1e3:8.2f} min={float(r['MinNs'])/1e3:7.2f} "
          f"max={float(r['MaxNs'])/1e3:7.2f} us  {float(r['Percentage']):5.1f}%")
