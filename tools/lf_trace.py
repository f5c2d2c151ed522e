#!/usr/bin/env python3
"""Timeline of workgroup 0 of the fused lookup + convc1 kernel (tuning build with -DMFTX_LF_TRACE:
tools/build_tuning.sh -DMFTX_LF_TRACE; MFTX_LIB=build_tune/libmftx_tune.so).  Prints, per wave, the events in
microseconds (s_memtime ticks at 100 MHz) relative to the workgroup's first stamp."""
import ctypes as C
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd import _lib, ops  # noqa: E402

NAMES = {1: "start", 2: "coords0", 3: "gather>", 4: "waited", 5: "convert>", 6: "bar<", 7: "bar>", 8: "mfma>", 9: "epi>",
         10: "table>", 11: "waited", 12: "taps>"}
P, h, w = 7, 64, 64
dev = "cuda"
g = torch.Generator().manual_seed(0)
N = h * w
f1 = torch.randn(P, N, 256, generator=g).to(dev)
f2 = torch.randn(P, N, 256, generator=g).to(dev)
lv = ops.corr_pyramid(f1, f2, h, w, arith=ops.ARITH_SPLIT)
ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
coords = (torch.stack([xs, ys], -1).reshape(1, N, 2).float() + 3 * torch.randn(P, N, 2, generator=g)).to(dev).contiguous()
wpk = ops.pack_conv_weight((torch.randn(256, 324, 1, 1, generator=g) * 0.05).to(dev))
bias = torch.randn(256, generator=g).to(dev)
wf = ops.pack_lookup_convc1_weights(wpk)
lib = _lib.load()
fn = lib.mftx_debug_lf_trace
fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * (8 * 128))()
flush = torch.empty(64 << 20, dtype=torch.float32, device=dev)
for rep in range(3):
    if os.environ.get("LF_TRACE_COLD"):
        flush.fill_(1.0)
    ops.corr_lookup_convc1(lv, coords, h, w, wf, bias, out_split=True)
    torch.cuda.synchronize()
    assert fn(buf) == 0
ev = [[(buf[wv * 128 + i] >> 56, buf[wv * 128 + i] & ((1 << 56) - 1)) for i in range(128) if buf[wv * 128 + i]] for wv in range(8)]
t0 = min(t for e in ev for _, t in e)
# (ticks are shader-clock cycles here: ~2.2 GHz)
for wv in (0, 4):
    print(f"wave {wv} ({'consumer' if wv < 4 else 'producer'}), kilo-cycles:")
    print("   " + "  ".join(f"{NAMES.get(c, c)}@{(t - t0) / 1000:.1f}" for c, t in ev[wv]))
prod = ev[4]
for name, code in (("gather", 3), ("table (bar> .. table>)", 10), ("gather wait (table> .. waited)", 11), ("tap reads (waited .. taps>)", 12),
                   ("DMA issue + conversion (taps> .. convert>)", 5)):
    d = [prod[i][1] - prod[i - 1][1] for i in range(1, len(prod)) if prod[i][0] == code]
    if d:
        print(f"producer {name}: median {sorted(d)[len(d) // 2]} cycles over {len(d)}")
cons = ev[0]
d = [cons[i][1] - cons[i - 1][1] for i in range(1, len(cons)) if cons[i][0] == 8]
print(f"consumer unit (72 MFMAs): median {sorted(d)[len(d) // 2]} cycles; total {(max(t for _, t in cons) - t0) / 1000:.1f} kilo-cycles")
