#!/usr/bin/env python3
"""Timeline of workgroup 0 of the tile-resident conv kernel (tuning build with -DMFTX_LF_TRACE): per wave, the phase
boundaries in kilo-ticks of s_memtime since the workgroup's first stamp, and the launch's duration.

    MFTX_LIB=build_tune/libmftx_tune.so python tools/tc_trace.py [cin cout kh kw]"""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd import _lib, ops  # noqa: E402

NAMES = {1: "start", 2: "loaded", 3: "bar>", 4: "mfma>", 5: "bar>", 6: "parked", 7: "bar>", 8: "stored"}
cin, cout, kh, kw = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (128, 256, 3, 3)
P, h, w = (int(sys.argv[5]) if len(sys.argv) > 5 else 7), 64, 64
M = P * h * w
g = torch.Generator().manual_seed(0)
xs = ops.split_activations(torch.randn(M, cin, generator=g).cuda())
x1, x2 = (xs, None) if cin == 128 else (xs[:, :128].contiguous(), xs[:, 128:].contiguous())
wpk = ops.pack_conv_weight((torch.randn(cout, cin, kh, kw, generator=g) * 0.05).cuda())
wtile = ops.pack_tile_conv_weights(wpk, cout, cin)
b = torch.randn(cout, generator=g).cuda()
lib = _lib.load()
fn = lib.mftx_debug_tc_trace
fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * (8 * 16))()
for rep in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.tile_conv2d(x1, wtile, b, P, h, w, cout, kh, kw, act="relu", x2=x2)
    e1.record()
    torch.cuda.synchronize()
    assert fn(buf) == 0
print(f"launch: {e0.elapsed_time(e1) * 1e3:.1f} us")
ev = [[(buf[wv * 16 + i] >> 56, buf[wv * 16 + i] & ((1 << 56) - 1)) for i in range(16) if buf[wv * 16 + i]] for wv in range(8)]
t0 = min(t for e in ev for _, t in e)
for wv in range(8):
    print(f"wave {wv}: " + "  ".join(f"{NAMES.get(c, c)}@{(t - t0) / 1000:.2f}" for c, t in ev[wv]))
