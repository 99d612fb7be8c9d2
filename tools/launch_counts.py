#!/usr/bin/env python
"""Launches per step by kernel, from a rocprofv3 --kernel-trace --stats output directory."""
import csv, glob, sys
d, steps = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(glob.glob(f"{d}/*/*_kernel_stats.csv")[0])))
print("launches/step", sum(int(r["Calls"]) for r in rows) / steps, "kernel ms/step",
      sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / steps)
rows.sort(key=lambda r: -int(r["Calls"]))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print("%7.1f  %6.1f us  %s" % (int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, r["Name"][:100]))
