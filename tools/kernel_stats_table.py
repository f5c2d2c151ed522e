#!/usr/bin/env python3
"""Top rows of a rocprofv3 kernel_stats.csv (run on the GPU box): python tools/kernel_stats_table.py <csv> [n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    name = r["Name"].replace("void ", "").replace("mftx::", "").split("(")[0][:60]
    print(f"{name:60s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:8.1f} us {100 * float(r['TotalDurationNs']) / tot:5.1f} %")
