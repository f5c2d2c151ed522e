#!/usr/bin/env python3
"""The fused flow branch (convf1 -> convf2 as one kernel, mftx_flow_branch) against convf2 alone as a split GEMM on a
materialised convf1 output (the stand-alone convf1 is a VALU kernel inside the engine: 27-28 us at 7 x 64 x 64,
bench.py `kernels.convf1`).

    python tools/bench_flow_branch.py [P h w]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd import ops  # noqa: E402

P, h, w = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (7, 64, 64)
M = P * h * w
g = torch.Generator().manual_seed(0)
w1 = (torch.randn(128, 2, 7, 7, generator=g) * 0.1).cuda()
w2 = (torch.randn(64, 128, 3, 3, generator=g) * 0.05).cuda()
b1, b2 = torch.randn(128, generator=g).cuda() * 0.1, torch.randn(64, generator=g).cuda() * 0.1
ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
coords = (torch.stack([xs, ys], -1).reshape(1, h * w, 2) + 3 * torch.randn(P, h * w, 2, generator=g)).cuda()
w2pk = ops.pack_conv_weight(w2)
wflow = ops.pack_flow_branch_weights(w1.permute(2, 3, 1, 0).reshape(98, 128).contiguous(), w2pk)
w2s = ops.split_weights(w2pk)
flo1 = ops.split_activations(torch.relu(torch.randn(M, 128, generator=g)).cuda())
corflo = torch.zeros(M, 256, device="cuda")
hx = torch.zeros(M, 384, device="cuda")
f2out = torch.zeros(M, 64, device="cuda")
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        flush.zero_()                                 # operands out of L2 / MALL, as inside the step
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


t_f = timed(lambda: ops.flow_branch(coords, h, w, wflow, b1, b2, out=corflo[:, 192:], hx=hx))
t_2 = timed(lambda: ops.conv2d(flo1, w2s, b2, P, h, w, 64, 3, 3, act="relu", arith=1, a_split=True, out_split=True, out=f2out))
flops = 2.0 * M * (98 * 128 + 1152 * 64)
print(f"P={P} {h}x{w}: fused flow branch {t_f:.1f} us ({3 * flops / t_f * 1e-6:.0f} TF of fp16 MFMA work) | convf2 alone as a split GEMM {t_2:.1f} us")
