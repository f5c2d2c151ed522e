#!/usr/bin/env python3
"""Per-kernel mean of every counter in a rocprofv3 counter_collection CSV (run on the GPU box):

    python tools/pmc_kernel_table.py <dir> [kernel substring]
"""
import collections
import csv
import sys
from pathlib import Path

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in Path(sys.argv[1]).rglob("*counter_collection.csv"):
    with open(p) as f:
        for r in csv.DictReader(f):
            if len(sys.argv) > 2 and sys.argv[2] not in r["Kernel_Name"]:
                continue
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        v = sorted(v)[len(v) // 2:]          # steady state: upper half
        print(f"    {c:32s} {sum(v) / len(v):16.1f}   (n={len(v)})")
