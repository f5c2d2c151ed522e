#!/usr/bin/env python3
"""Timeline of workgroup 0 of the fused flow-branch kernel (tuning build with -DMFTX_LF_TRACE: tools/build_tuning.sh
-DMFTX_LF_TRACE; MFTX_LIB=build_tune/libmftx_tune.so): per wave, the stage boundaries in kilo-cycles since the
workgroup's first stamp.

    MFTX_LIB=build_tune/libmftx_tune.so python tools/fb_trace.py [P h w]"""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd import _lib, ops  # noqa: E402

NAMES = {1: "start", 2: "flow>", 3: "bar>", 4: "mfma1>", 5: "epi1>", 6: "bar>", 7: "mfma2>", 8: "red>", 9: "bar>", 10: "stored"}
P, h, w = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (7, 64, 64)
g = torch.Generator().manual_seed(0)
w1 = (torch.randn(128, 2, 7, 7, generator=g) * 0.1).cuda()
w2 = (torch.randn(64, 128, 3, 3, generator=g) * 0.05).cuda()
b1, b2 = torch.randn(128, generator=g).cuda() * 0.1, torch.randn(64, generator=g).cuda() * 0.1
ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
coords = (torch.stack([xs, ys], -1).reshape(1, h * w, 2) + 3 * torch.randn(P, h * w, 2, generator=g)).cuda()
wflow = ops.pack_flow_branch_weights(w1.permute(2, 3, 1, 0).reshape(98, 128).contiguous(), ops.pack_conv_weight(w2))
lib = _lib.load()
fn = lib.mftx_debug_fb_trace
fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * (8 * 16))()
for rep in range(3):
    ops.flow_branch(coords, h, w, wflow, b1, b2)
    torch.cuda.synchronize()
    assert fn(buf) == 0
ev = [[(buf[wv * 16 + i] >> 56, buf[wv * 16 + i] & ((1 << 56) - 1)) for i in range(16) if buf[wv * 16 + i]] for wv in range(8)]
t0 = min(t for e in ev for _, t in e)
for wv in range(8):
    print(f"wave {wv}: " + "  ".join(f"{NAMES.get(c, c)}@{(t - t0) / 1000:.2f}" for c, t in ev[wv]))
