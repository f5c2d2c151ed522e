#!/usr/bin/env python3
"""Micro-benchmark of the small-N 3x3 conv (flow head's last layer, 256 -> 2)."""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--P", type=int, default=7)
ap.add_argument("--h", type=int, default=64)
ap.add_argument("--w", type=int, default=64)
a = ap.parse_args()
M = a.P * a.h * a.w
x = torch.randn(M, 256, device="cuda")
wt = ops.pack_conv_weight(torch.randn(2, 256, 3, 3, device="cuda") * 0.05)
b = torch.randn(2, device="cuda")
for _ in range(3):
    ops.conv2d(x, wt, b, a.P, a.h, a.w, 2, 3, 3)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    ops.conv2d(x, wt, b, a.P, a.h, a.w, 2, 3, 3)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 50 * 1e-3
print(f"P={a.P} {a.h}x{a.w}: {t * 1e6:.1f} us ({M * 1024 / t / 1e9:.0f} GB/s of input map)")
