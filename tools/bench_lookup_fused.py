#!/usr/bin/env python3
"""Micro-benchmark of the lookup fused into convc1 (mftx_corr_lookup_convc1) against the two kernels it replaces, on a
cold-ish pyramid (a 1 GB buffer is rewritten between launches).  With a tuning build (tools/build_tuning.sh,
MFTX_LIB=build_tune/libmftx_tune.so) MFTX_LF_ABLATE removes parts of the fused kernel (results are then garbage).
    python tools/bench_lookup_fused.py [P] [h] [w]"""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd import ops  # noqa: E402

P, h, w = (int(a) for a in (sys.argv[1:4] + ["7", "64", "64"][len(sys.argv) - 1:]))
dev = "cuda"
g = torch.Generator().manual_seed(0)
N = h * w
f1 = torch.randn(P, N, 256, generator=g).to(dev)
f2 = torch.randn(P, N, 256, generator=g).to(dev)
lv = ops.corr_pyramid(f1, f2, h, w, arith=ops.ARITH_SPLIT)
ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
grid = torch.stack([xs, ys], -1).reshape(1, N, 2).float()
coords = (grid + 3 * torch.randn(P, N, 2, generator=g)).to(dev).contiguous()
wpk = ops.pack_conv_weight((torch.randn(256, 324, 1, 1, generator=g) * 0.05).to(dev))
wsp = ops.split_weights(wpk)
bias = torch.randn(256, generator=g).to(dev)
wf = ops.pack_lookup_convc1_weights(wpk)
flush = torch.empty(256 << 20, dtype=torch.float32, device=dev)


def timeit(fn, n=20):
    ts = []
    for _ in range(n):
        flush.fill_(1.0)                       # evict the pyramid from the 256 MB MALL / L2
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


abl = os.environ.get("MFTX_LF_ABLATE", "0")
t_f = timeit(lambda: ops.corr_lookup_convc1(lv, coords, h, w, wf, bias, out_split=True))
t_l = timeit(lambda: ops.corr_lookup(lv, coords, h, w))
feats = ops.corr_lookup(lv, coords, h, w).reshape(P * N, 324)
t_c = timeit(lambda: ops.conv2d(feats, wsp, bias, P, h, w, 256, 1, 1, act="relu", arith=ops.ARITH_SPLIT, out_split=True))
alg = P * N * (1600 + 8 + 1296)
print(f"P={P} {h}x{w} ablate={abl}: fused {t_f:.1f} us | lookup {t_l:.1f} + convc1 {t_c:.1f} = {t_l + t_c:.1f} us | "
      f"lookup-algorithmic {alg / 1e6:.1f} MB: fused {alg / t_f / 1e6:.2f} TB/s, fused - convc1 {alg / max(t_f - t_c, 1e-3) / 1e6:.2f} TB/s")
