#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 kernel trace (run on the GPU box):

    python tools/trace_gaps.py <dir with *_kernel_trace.csv> [last N kernels]
"""
import csv
import sys
from pathlib import Path

rows = []
for p in Path(sys.argv[1]).rglob("*kernel_trace.csv"):
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
if len(sys.argv) > 2:
    rows = rows[-int(sys.argv[2]):]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = [max(0, rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1)]
gaps_small = [g for g in gaps if g < 50000]          # < 50 us: back-to-back launches, not host pauses
print(f"{len(rows)} kernels, span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms ({100 * busy / span:.1f} %)")
print(f"gaps < 50 us: n={len(gaps_small)}, total {sum(gaps_small) / 1e6:.3f} ms, mean {sum(gaps_small) / max(1, len(gaps_small)) / 1e3:.2f} us, "
      f"median {sorted(gaps_small)[len(gaps_small) // 2] / 1e3:.2f} us")
print(f"gaps >= 50 us: n={len(gaps) - len(gaps_small)}, total {(sum(gaps) - sum(gaps_small)) / 1e6:.3f} ms")
