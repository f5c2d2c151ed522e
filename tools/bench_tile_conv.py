#!/usr/bin/env python3
"""The tile-resident conv kernel (mftx_tile_conv2d) against the ring-buffered GEMM (mftx_conv2d) on the layers it covers,
operands evicted from L2 / MALL before every launch (as inside the step).

    python tools/bench_tile_conv.py [P h w]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd import ops  # noqa: E402

P, h, w = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (7, 64, 64)
M = P * h * w
g = torch.Generator().manual_seed(0)
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")


def timed(fn, reps=15):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


tot_t = tot_r = 0.0
for name, cin, cout, kh, kw, per_frame in (("fh1 / mask0 3x3 128->256", 128, 256, 3, 3, 13), ("gru zr 1x5 256->256", 256, 256, 1, 5, 12),
                                           ("gru zr 5x1 256->256", 256, 256, 5, 1, 12), ("gru q 1x5 256->128", 256, 128, 1, 5, 12),
                                           ("gru q 5x1 256->128", 256, 128, 5, 1, 12), ("gru inp 1x5 128->256", 128, 256, 1, 5, 1),
                                           ("gru inp 5x1 128->128", 128, 128, 5, 1, 1)):
    xs = ops.split_activations(torch.randn(M, cin, generator=g).cuda())
    x1, x2 = (xs, None) if cin == 128 else (xs[:, :128].contiguous(), xs[:, 128:].contiguous())
    wpk = ops.pack_conv_weight((torch.randn(cout, cin, kh, kw, generator=g) * 0.05).cuda())
    wsp, wtile = ops.split_weights(wpk), ops.pack_tile_conv_weights(wpk, cout, cin)
    b = torch.randn(cout, generator=g).cuda()
    out = torch.empty(M, cout, device="cuda")
    t_t = timed(lambda: ops.tile_conv2d(x1, wtile, b, P, h, w, cout, kh, kw, act="relu", x2=x2, out_split=True, out=out))
    t_r = timed(lambda: ops.conv2d(x1, wsp, b, P, h, w, cout, kh, kw, act="relu", x2=x2, arith=1, a_split=True, out_split=True, out=out))
    gf = 2.0 * M * cout * kh * kw * cin * 1e-9
    tot_t += per_frame * t_t
    tot_r += per_frame * t_r
    print(f"{name:28s} tile-resident {t_t:6.1f} us ({gf / t_t * 1e3:5.0f} TF alg.) | ring GEMM {t_r:6.1f} us ({gf / t_r * 1e3:5.0f} TF alg.)")
print(f"per frame (12 iterations): tile-resident {tot_t * 1e-3:.2f} ms | ring GEMM {tot_r * 1e-3:.2f} ms")
