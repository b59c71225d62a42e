#!/usr/bin/env python3
"""From a rocprofv3 kernel_trace.csv: launches, summed kernel time and first-start-to-last-end span of the LAST n kernels (default 1000)."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rows = rows[-n:]
dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e3
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print(f"{len(rows)} kernels: sum of durations {dur:.1f} us, span {span:.1f} us, GPU busy {dur / span:.2f}")
