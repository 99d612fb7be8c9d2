#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats output directory: top kernels per step."""
import csv, glob, sys, collections
d = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 7; top = int(sys.argv[3]) if len(sys.argv) > 3 else 28
f = (glob.glob(f'{d}/*/*_kernel_stats.csv') + glob.glob(f'{d}/*_kernel_stats.csv'))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'total kernel ms/step {tot/1e6/steps:.2f}')
for r in rows[:top]:
    print(f"{r['Name'][:86]:86s} n/step={int(r['Calls'])/steps:6.1f} ms/step={float(r['TotalDurationNs'])/1e6/steps:6.2f} avg_us={float(r['AverageNs'])/1e3:8.1f} max_us={float(r['MaxNs'])/1e3:8.1f}")
