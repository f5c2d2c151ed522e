#!/usr/bin/env python3
"""Thread sweep of the CPU oracle (bench.py's `cpu_baseline` leg): one 512x512 / 12-iteration flow pair
(oracle/mft_oracle.py, torch CPU ops) at 8 .. all host cores.  Run once on the GPU box's host; the result
(profiles/r2_cpu_thread_sweep.txt) justifies bench.py's default of min(cores, 32) threads.

    python tools/cpu_thread_sweep.py [--size 512] [--iters 12]
"""
import argparse
import os
import sys
import time
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from oracle import mft_oracle as O  # noqa: E402  (test infrastructure; this tool times it, nothing ships it)
from mft_amd.synth import SyntheticVideo  # noqa: E402
from mft_amd.weights import make_weights  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=512)
ap.add_argument("--iters", type=int, default=12)
a = ap.parse_args()
cores = len(os.sched_getaffinity(0))
sd = {k: torch.from_numpy(v) for k, v in make_weights(0).items()}
vid = SyntheticVideo(a.size, a.size, n_frames=8, seed=0)
print(f"# host cores {cores}; one {a.size}x{a.size} pair, {a.iters} iterations, oracle.compute_flow")
ap2 = [t for t in (8, 16, 32, 64, 128) if t <= cores]      # (256 threads on a 256-core host took 783 s per pair: oversubscribed, dropped)
for n in ap2:
    torch.set_num_threads(n)
    with torch.no_grad():
        O.compute_flow(sd, vid[0], vid[1], 1)                      # touch everything once
        t0 = time.perf_counter()
        O.compute_flow(sd, vid[0], vid[4], a.iters)
        dt = time.perf_counter() - t0
    print(f"threads {n:4d}: {dt:6.2f} s per pair  ->  {7 * dt:6.1f} s per 7-pair frame")
