#!/usr/bin/env python
"""Print the kernels launched around the longest instances of kernels matching a pattern (rocprofv3 --kernel-trace csv)."""
import csv, glob, sys
d, pat = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 3
rows = list(csv.DictReader(open(glob.glob(f"{d}/*/*_kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
hits = sorted([i for i, r in enumerate(rows) if pat in r["Kernel_Name"]], key=lambda i: -dur(rows[i]))[:top]
for i in hits:
    print(f"--- {rows[i]['Kernel_Name'][:70]} {dur(rows[i]):.1f} us")
    for j in range(max(0, i - 4), min(len(rows), i + 4)):
        print(f"   {'>>' if j == i else '  '} {dur(rows[j]):8.1f} us  {rows[j]['Kernel_Name'][:100]}")
