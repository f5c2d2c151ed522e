#!/usr/bin/env python3
"""What a plain device-to-device copy achieves at the byte counts of the small HBM-bound kernels (one launch per
measurement, HIP events, like bench.py brackets them): the practical ceiling for a ~20 us kernel, where launch, ramp-up and
drain are a fixed ~5 us.  Run on the GPU box."""
import torch

for total_mb in (62.9, 96.0, 111.0, 250.0, 501.0, 2000.0):
    n = int(total_mb * 1e6 / 2 / 4)                 # floats per side: half the traffic is read, half written
    a = torch.empty(n, device="cuda"); b = torch.randn(n, device="cuda")
    for _ in range(3):
        a.copy_(b)
    torch.cuda.synchronize()
    ts = []
    junk = torch.randn(64 << 20, device="cuda")     # 256 MB: evict the MALL between measurements
    for _ in range(10):
        junk.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); a.copy_(b); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    t = sorted(ts)[len(ts) // 2]
    print(f"copy with {total_mb:7.1f} MB of traffic: {t:7.1f} us  {total_mb * 1e6 / t / 1e6:6.2f} TB/s = {total_mb * 1e6 / t / 1e6 / 8:.2f} of 8 TB/s")
