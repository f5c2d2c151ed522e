#!/usr/bin/env python3
"""Do kernels of different streams ever run at the same time?  From a rocprofv3 --kernel-trace CSV (second half = steady state): wall
time, the sum of the kernel durations, the time covered by at least one kernel, kernels per hardware queue, idle gaps > 50 us.
Sum = covered means NO two kernels overlapped -- how HIP's default of four hardware queues was found to serialise the sharded
per-frame mode (profiles/r5s_hw_queues.txt).

    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python bench.py ...;  python tools/trace_overlap.py out/.../t_kernel_trace.csv"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(rows); rows = rows[n // 2:]            # steady part
t0 = int(rows[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in rows)
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
# union of intervals
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows)
cov = 0; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce: cov += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
cov += ce - cs
q = collections.Counter(r.get('Queue_Id', '?') for r in rows)
print(f"wall {(t1 - t0) / 1e6:.2f} ms, sum of kernel times {busy / 1e6:.2f} ms, covered {cov / 1e6:.2f} ms, queues {dict(q)}")
# idle gaps > 50 us
gaps = []
cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce:
        if s - ce > 50000: gaps.append(((ce - t0) / 1e6, (s - ce) / 1e3))
        cs, ce = s, e
    else: ce = max(ce, e)
print("idle gaps > 50 us (at ms, us):", [(round(a, 2), round(b)) for a, b in gaps[:20]], len(gaps))
