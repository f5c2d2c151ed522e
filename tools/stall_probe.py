#!/usr/bin/env python3
"""Per-call wall time of `compute_flow_many` at P = 7 over many calls: looks for sporadic host / runtime stalls (a call above 1.5 x the
median is listed with its host enqueue time, the allocator's growth and the GC counters).

    python tools/stall_probe.py [calls]      # round 4: 3000 calls, median 6.20 ms, mean 6.206, max 7.15 -- none
"""
import sys, time, gc, torch
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from mft_amd.config import load_config
from mft_amd.synth import SyntheticVideo
conf = load_config(REPO / "configs" / "MFT_cfg.py")
fc = conf.flow_config
fc.model = None; fc.synthetic_weights_seed = 0; fc.flow_iters = 12; fc.async_encode = False; fc.split_streams = 1
flower = fc.of_class(fc)
vid = SyntheticVideo(512, 512, n_frames=8, seed=0)
frames = [vid[i] for i in range(8)]
P = 7
lefts = [(i, frames[i]) for i in range(P)]
right = (7, frames[7])
for _ in range(5):
    flower.compute_flow_many(lefts, right)
torch.cuda.synchronize()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
ts = []
t_start = time.perf_counter()
for i in range(N):
    r0 = torch.cuda.memory_reserved()
    g0 = gc.get_count()
    t0 = time.perf_counter()
    flower.compute_flow_many(lefts, right)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append(((t2 - t0) * 1e3, (t1 - t0) * 1e3, t0 - t_start, torch.cuda.memory_reserved() - r0, g0))
import statistics
med = statistics.median(t[0] for t in ts)
print(f"{N} calls at P={P}: median {med:.2f} ms, mean {sum(t[0] for t in ts) / N:.3f} ms, max {max(t[0] for t in ts):.2f} ms")
for i, t in enumerate(ts):
    if t[0] > 1.5 * med:
        print(f"  call {i} at {t[2]:.2f} s: total {t[0]:.2f} ms, host enqueue {t[1]:.2f} ms, reserved delta {t[3]}, gc counts {t[4]}")
print("graphs", flower.engine.graph_stats())
