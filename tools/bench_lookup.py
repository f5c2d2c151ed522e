#!/usr/bin/env python3
"""Micro-benchmark of mftx_corr_lookup at the tracker's batch sizes."""
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
args = ap.parse_args()
P, h, w = args.P, args.h, args.w
N = h * w
stride, _ = ops.pyramid_layout(h, w)
lv = [torch.randn(P, N, stride[l], device="cuda") for l in range(4)]       # stored layout (padding content is never read)
ys, xs = torch.meshgrid(torch.arange(h, device="cuda"), torch.arange(w, device="cuda"), indexing="ij")
coords = (torch.stack([xs, ys], -1).reshape(1, N, 2).float() + torch.randn(P, N, 2, device="cuda")).contiguous()
for _ in range(3):
    ops.corr_lookup(lv, coords, h, w)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    ops.corr_lookup(lv, coords, h, w)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 50 * 1e-3
b = P * N * (1600 + 8 + 1296)
print(f"P={P} {h}x{w}: {t * 1e6:.1f} us, {b / t / 1e9:.0f} GB/s algorithmic ({b / 1e6:.1f} MB)")
