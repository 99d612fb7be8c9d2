#!/usr/bin/env python
"""Print the kernel sequence of the last complete training step of a rocprofv3 --kernel-trace csv (steps are delimited
by the k_adam launch), one line per launch: duration and name."""
import csv, glob, re, sys
d = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(f"{d}/*/*_kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "k_adam" in r["Kernel_Name"]]
a, b = adam[-2] + 1, adam[-1] + 1
for r in rows[a:b]:
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print(f"{us:9.1f}  {re.sub(r'^void ', '', r['Kernel_Name'])[:110]}")
