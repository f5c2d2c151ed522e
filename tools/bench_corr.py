#!/usr/bin/env python3
"""Micro-benchmark of mftx_corr_pyramid (volume GEMM + pooling)."""
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
ap.add_argument("--arith", type=int, default=1, help="0 fp32 MFMA, 1 split fp16 (the engine's default)")
a = ap.parse_args()
N = a.h * a.w
f1 = torch.randn(a.P, N, 256, device="cuda")
f2 = torch.randn(a.P, N, 256, device="cuda")
for _ in range(3):
    ops.corr_pyramid(f1, f2, a.h, a.w, arith=a.arith)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.corr_pyramid(f1, f2, a.h, a.w, arith=a.arith)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20 * 1e-3
fl = 2.0 * a.P * N * N * 256
print(f"P={a.P} {a.h}x{a.w} arith {a.arith}: volume+pool {t * 1e6:.1f} us  ({fl / t / 1e12:.1f} TFLOP/s if it were all GEMM)")
