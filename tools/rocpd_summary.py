#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (``--kernel-trace --stats`` output) as a
per-kernel table: calls, total / average duration, share of GPU time.

    python tools/rocpd_summary.py gpurun_out/prof/bench_results.db > profiles/xyz.txt
"""
import sqlite3
import sys


def short(name, n=110):
    name = name.replace("mftx::", "mftx::")
    return name if len(name) <= n else name[: n - 3] + "..."


def main(path):
    db = sqlite3.connect(path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    tot = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print(f"# total kernel time {tot / 1e3:.1f} ms over {sum(r[1] for r in rows)} dispatches (durations in us)")
    print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'pct':>6}  kernel")
    for name, calls, total, avg, pct in rows[:40]:
        print(f"{calls:7d} {total:12.1f} {avg:10.2f} {pct:6.2f}  {short(name)}")
    mine = [r for r in rows if r[0].startswith("void mftx::") or r[0].startswith("mftx::")]
    print(f"# libmftx kernels: {sum(r[2] for r in mine) / 1e3:.1f} ms = {100 * sum(r[2] for r in mine) / tot:.1f}% of GPU time")
    q = ("select name, count(*), avg(duration)/1e3, min(duration)/1e3, max(duration)/1e3, avg(vgpr_count), "
         "avg(accum_vgpr_count), avg(lds_size) from kernels where name like '%mftx::%' group by name order by sum(duration) desc")
    print("# per-kernel resources (avg_us, min_us, max_us, vgpr, agpr, lds_bytes):")
    for r in db.execute(q):
        print(f"#   {short(r[0], 70):70s} n={r[1]:5d} avg={r[2]:9.2f} min={r[3]:9.2f} max={r[4]:9.2f} vgpr={r[5]:.0f} agpr={r[6]:.0f} lds={r[7]:.0f}")


if __name__ == "__main__":
    main(sys.argv[1])
