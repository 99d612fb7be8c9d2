#!/usr/bin/env python
"""Per-kernel averages of every counter found under a directory of rocprofv3 --pmc passes (csv output)."""
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:34s} n={len(v):3d} avg={sum(v)/len(v):16.1f}")
