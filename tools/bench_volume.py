#!/usr/bin/env python3
"""The correlation-volume GEMM (+ pooled levels) alone: python tools/bench_volume.py [P h w]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd import ops  # noqa: E402

P, h, w = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (7, 64, 64)
g = torch.Generator().manual_seed(0)
f1 = torch.randn(P, h * w, 256, generator=g).cuda()
f2 = torch.randn(P, h * w, 256, generator=g).cuda()
for _ in range(2):
    lv = ops.corr_pyramid(f1, f2, h, w, arith=1)
torch.cuda.synchronize()
reps = 5
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    lv = ops.corr_pyramid(f1, f2, h, w, arith=1)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / reps * 1e-3
N = h * w
gb = sum(x.numel() for x in lv) * 4e-9
print(f"P={P} {h}x{w}: {t * 1e6:.1f} us, {3 * 2.0 * P * N * N * 256 / t * 1e-12:.0f} TF of fp16 MFMA work, {gb / t * 1e-3:.2f} TB/s of pyramid written ({gb:.2f} GB)")
