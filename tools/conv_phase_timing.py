#!/usr/bin/env python3
"""Per-phase cycle totals of the split-arithmetic K loop (tuning build -DMFTX_TIMING, selected with MFTX_LIB):
    bash tools/build_ablations.sh T ; MFTX_LIB=mft_amd/csrc/abl/libmftx_T.so python tools/conv_phase_timing.py [tile]
phases per chunk: group 0 | wait LDS | wait DMA | barrier | LDS reads + refill issue | group 1"""
import ctypes as C
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd import _lib, ops  # noqa: E402

lib = _lib.load()
raw = C.CDLL(os.environ["MFTX_LIB"])
buf = (C.c_ulonglong * 16)()
P, h, w = 7, 64, 64
M = P * h * w
PRE = "--a-split" in sys.argv        # A and output in split form (what the engine runs)
TILE = int(sys.argv[sys.argv.index("--tile") + 1]) if "--tile" in sys.argv else None
for name, cin, cout, kh, kw in (("gru zr 1x5 256->256", 256, 256, 1, 5), ("gru q 1x5 256->128", 256, 128, 1, 5),
                                ("fh1 3x3 128->256", 128, 256, 3, 3), ("convc1 1x1 324->256", 324, 256, 1, 1),
                                ("convc2 3x3 256->192", 256, 192, 3, 3), ("convf2 3x3 128->64", 128, 64, 3, 3)):
    x = torch.randn(M, cin, device="cuda")
    if PRE:
        x = ops.split_activations(x)
    kw_ = dict(a_split=PRE, out_split=PRE and cout % 8 == 0, tile=TILE)
    wt = ops.split_weights(ops.pack_conv_weight(torch.randn(cout, cin, kh, kw, device="cuda") * 0.05))
    b = torch.randn(cout, device="cuda")
    for _ in range(3):
        ops.conv2d(x, wt, b, P, h, w, cout, kh, kw, act="relu", arith=1, **kw_)
    torch.cuda.synchronize()
    raw.mftx_debug_timing(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 10
    for _ in range(reps):
        ops.conv2d(x, wt, b, P, h, w, cout, kh, kw, act="relu", arith=1, **kw_)
    e1.record()
    torch.cuda.synchronize()
    raw.mftx_debug_timing(buf, 1)
    n = max(buf[6], 1)
    names = ("group0", "waitLDS", "waitDMA", "barrier", "reads+refill", "group1")
    tot = sum(buf[i] for i in range(6))
    nt = max(buf[11], 1)
    print(f"{name}: per wave-tile (ticks): prologue {buf[8] / nt:.0f}, K loop {buf[9] / nt:.0f}, epilogue (stores issued) {buf[10] / nt:.0f}; wave-tiles per launch {nt / reps:.0f}")
    print(f"{name}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us; per wave-chunk {tot / n:.0f} ticks (100 MHz ticks x 24 = cycles at 2.4 GHz?): "
          + ", ".join(f"{nm} {buf[i] / n:.1f}" for i, nm in enumerate(names)))

# the correlation volume (split arithmetic): tile phases per wave -- prologue (first two chunks in flight -> landed), K loop
# (8 chunks), epilogue (stage through LDS, four levels stored, stores drained)
f1 = torch.randn(P, h * w, 256, device="cuda")
f2 = torch.randn(P, h * w, 256, device="cuda")
for _ in range(3):
    ops.corr_pyramid(f1, f2, h, w, arith=1)
torch.cuda.synchronize()
raw.mftx_debug_timing(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.corr_pyramid(f1, f2, h, w, arith=1)
e1.record()
torch.cuda.synchronize()
raw.mftx_debug_timing(buf, 1)
n = max(buf[11], 1)
print(f"corr volume 7 x 4096 x 4096: {e0.elapsed_time(e1) / reps * 1e3:.1f} us; per wave-tile (ticks): prologue {buf[8] / n:.0f}, K loop {buf[9] / n:.0f}, "
      f"epilogue {buf[10] / n:.0f}; wave-tiles per launch {n / reps:.0f}")
