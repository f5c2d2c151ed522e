#!/usr/bin/env python3
"""Timeline of workgroup 0 of the fused GRU-pass kernel (tuning build with -DMFTX_LF_TRACE): per wave, the phase boundaries in
kilo-cycles of s_memtime since the workgroup's first stamp, and every workgroup's measured clock.

    MFTX_LIB=build_tune/libmftx_tune.so python tools/gru_trace.py [P] [vertical]"""
import ctypes as C
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd import _lib, ops  # noqa: E402

NAMES = {1: "start", 2: "loaded", 3: "bar>", 4: "gates>", 5: "z,rh>", 6: "bar>", 7: "cand>", 8: "stored"}
P = int(sys.argv[1]) if len(sys.argv) > 1 else 7
vertical = len(sys.argv) > 2 and sys.argv[2] not in ("0", "")
h = w = 64
M = P * h * w
g = torch.Generator().manual_seed(0)
hf = torch.tanh(torch.randn(M, 128, generator=g)).cuda()
mo = torch.relu(torch.randn(M, 128, generator=g)).cuda()
kh, kw = (5, 1) if vertical else (1, 5)
pack = lambda n: ops.pack_tile_conv_weights(ops.pack_conv_weight((torch.randn(n, 256, kh, kw, generator=g) * 0.04).cuda()), n, 256)  # noqa: E731
wzr, wq = pack(256), pack(128)
pre_zr, pre_q = (torch.randn(M, 256, generator=g) * 0.5).cuda(), (torch.randn(M, 128, generator=g) * 0.5).cuda()
hs, ms = ops.split_activations(hf), ops.split_activations(mo)
lib = _lib.load()
ftrace, fclk = lib.mftx_debug_tc_trace, lib.mftx_debug_tc_clock
ftrace.restype, ftrace.argtypes = C.c_int, [C.POINTER(C.c_ulonglong)]
fclk.restype, fclk.argtypes = C.c_int, [C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * (8 * 16))()
for rep in range(30):
    if rep == 29:
        assert ftrace(buf) == 0          # (reading clears: the last launch's stamps follow)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    ops.gru_half(hs, ms, wzr, wq, pre_zr, pre_q, hf, P, h, w, vertical=vertical)
e1.record()
torch.cuda.synchronize()
assert ftrace(buf) == 0
n_wg = P * 32
cb = (C.c_ulonglong * (4 * n_wg))()
assert fclk(cb, n_wg) == 0
clk = [(cb[4 * i + 1] - cb[4 * i]) / max(cb[4 * i + 3] - cb[4 * i + 2], 1) * 0.1 for i in range(n_wg)]
cyc = [cb[4 * i + 1] - cb[4 * i] for i in range(n_wg)]
print(f"P = {P}, {'vertical' if vertical else 'horizontal'} pass: launch {e0.elapsed_time(e1) * 1e3:.1f} us (30th of a loop); {n_wg} workgroups: "
      f"{statistics.median(cyc) / 1e3:.1f} k cycles each at {statistics.median(clk):.2f} GHz")
ev = [[(buf[wv * 16 + i] >> 56, buf[wv * 16 + i] & ((1 << 56) - 1)) for i in range(16) if buf[wv * 16 + i]] for wv in range(8)]
t0 = min(t for e in ev for _, t in e)
for wv in (0, 3, 4, 7):
    print(f"wave {wv}: " + "  ".join(f"{NAMES.get(c, c)}@{(t - t0) / 1000:.2f}" for c, t in ev[wv]))
