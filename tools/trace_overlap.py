#!/usr/bin/env python3
"""Do kernels of different streams ever run at the same time?  From a rocprofv3 --kernel-trace CSV (second half = steady state): wall
time, the sum of the kernel durations, the time covered by at least one kernel, kernels per hardware queue, idle gaps > 50 us.
Sum = covered means NO two kernels overlapped.  CAUTION: under rocprofv3's kernel trace the tracker's two lanes do not overlap at all
(queue 3 runs its 98-kernel batch, then queue 4 its own), while un-profiled event timing of the same run shows two batches in flight
all the time (each ~10-12 ms long, one finishing every ~5.1 ms): the profiler serialises them.  Use this for gaps and per-queue busy
time, not to decide whether streams overlap -- that is what A/B runs are for (profiles/r5s_hw_queues.txt).

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

# per-queue busy time and pairwise overlap (time during which BOTH queues have a kernel running)
byq = collections.defaultdict(list)
for r in rows:
    byq[r.get('Queue_Id', '?')].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))


def union(iv):
    iv = sorted(iv); out = []; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: out.append((cs, ce)); cs, ce = s, e
        else: ce = max(ce, e)
    out.append((cs, ce)); return out


def inter(a, b):
    i = j = 0; t = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e: t += e - s
        if a[i][1] < b[j][1]: i += 1
        else: j += 1
    return t


u = {q: union(v) for q, v in byq.items()}
print("busy per queue (ms):", {q: round(sum(e - s for s, e in v) / 1e6, 1) for q, v in u.items()})
qs = sorted(u)
print("pairwise overlap (ms):", {f"{a}&{b}": round(inter(u[a], u[b]) / 1e6, 1) for i, a in enumerate(qs) for b in qs[i + 1:]})
names = {q: collections.Counter(re.sub(r'<.*', '', r['Kernel_Name'].replace('mftx::', ''))[:28] for r in rows if r.get('Queue_Id', '?') == q).most_common(3) for q in qs}
print("top kernels per queue:", names)
