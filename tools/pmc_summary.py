#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel over the passes found under a directory.

    python tools/pmc_summary.py gpurun_out/pmc_r2 [substring of the kernel name]
"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "lighting"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"{root}/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if want in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k)
    for c in sorted(v):
        x = v[c]
        print(f"   {c:36s} {sum(x) / len(x):16.1f}   (n={len(x)})")
