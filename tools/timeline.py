#!/usr/bin/env python3
"""Print the kernel timeline of the LAST training step found in a rocprofv3 kernel trace
(anchor: the last launch whose name contains argv[2], default k_tiny64)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_tiny64"
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
lo = max(0, idx[-1] - 8)
t0 = int(rows[idx[-1]]["Start_Timestamp"])
for r in rows[lo:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:100]}")
