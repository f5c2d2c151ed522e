#!/usr/bin/env python3
"""Per-layer micro-benchmark of the fp32-MFMA implicit-GEMM conv kernel
(`mftx_conv2d`) on the shapes of the RAFT update block / OU heads, at the
batch sizes the tracker uses (P = 7 pairs on one GPU, P = 1 when delta-sharded).

    python tools/bench_conv.py [--P 7] [--h 64] [--w 64]
"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd import ops  # noqa: E402

LAYERS = [
    # name, cin, cout, kh, kw, calls per iteration (12 iters) or per pair
    ("convc1 1x1 324->256", 324, 256, 1, 1, 12),
    ("convc2 3x3 256->192", 256, 192, 3, 3, 12),
    ("convf2 3x3 128->64", 128, 64, 3, 3, 12),
    ("conv   3x3 256->126", 256, 126, 3, 3, 12),
    # GRU gates: the inp third of the input is hoisted out of the loop -> 256 input channels
    ("gru zr 1x5 256->256", 256, 256, 1, 5, 12),
    ("gru q  1x5 256->128", 256, 128, 1, 5, 12),
    ("gru zr 5x1 256->256", 256, 256, 5, 1, 12),
    ("gru q  5x1 256->128", 256, 128, 5, 1, 12),
    ("gru inp 1x5 128->384", 128, 384, 1, 5, 1),
    ("gru inp 5x1 128->384", 128, 384, 5, 1, 1),
    ("fh1    3x3 128->256", 128, 256, 3, 3, 12),
    ("fh2    3x3 256->2", 256, 2, 3, 3, 12),
    ("mask0  3x3 128->256", 128, 256, 3, 3, 1),
    ("mask2  1x1 256->576", 256, 576, 1, 1, 1),
    ("ou1    3x3 712->256", 712, 256, 3, 3, 1),
    ("ou2    3x3 256->3", 256, 3, 3, 3, 1),
    # one K chunk: the fixed cost of a launch + one round of tiles (prologue, pipeline fill, epilogue); not in the total
    ("tiny   1x1 32->256", 32, 256, 1, 1, 0),
    ("tiny   1x1 32->128", 32, 128, 1, 1, 0),
    ("tiny   1x1 128->256", 128, 256, 1, 1, 0),
]


# encoder layers (core/extractor.py), stride-1 ones: (name, cin, cout, k, grid)
ENC_LAYERS = [
    ("enc l1 3x3 64->64 @256", 64, 64, 3, 256),
    ("enc l2 3x3 96->96 @128", 96, 96, 3, 128),
    ("enc l3 3x3 128->128 @64", 128, 128, 3, 64),
    ("enc head 1x1 128->256 @64", 128, 256, 1, 64),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--P", type=int, default=7)
    ap.add_argument("--h", type=int, default=64)
    ap.add_argument("--w", type=int, default=64)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default=None, help="substring of the layer name")
    ap.add_argument("--arith", type=int, default=0, help="0 fp32 MFMA, 1 split fp16")
    ap.add_argument("--a-split", action="store_true", help="A operand pre-split (split arithmetic)")
    ap.add_argument("--out-split", action="store_true", help="output written in split form")
    ap.add_argument("--enc", action="store_true", help="the encoder's layer shapes (one 512 x 512 frame) instead")
    ap.add_argument("--tile", type=int, default=None, help="force the workgroup tile shape (mftx_conv2d_tile)")
    args = ap.parse_args()
    P, h, w = args.P, args.h, args.w
    M = P * h * w
    dev = "cuda"
    tot_t = tot_f = 0.0
    print(f"M = {M} cells (P={P}, {h}x{w})")
    layers = LAYERS
    if args.enc:
        layers = [(n, ci, co, k, k, 1, g) for n, ci, co, k, g in ENC_LAYERS]
    for layer in layers:
        name, cin, cout, kh, kw, calls = layer[:6]
        if args.enc:
            P, h, w = 1, layer[6], layer[6]
            M = h * w
        if args.only and args.only not in name:
            continue
        cin_s = -(-cin // 8) * 8 if args.a_split else cin       # split form: whole 8-channel groups (324 -> 328)
        x = torch.randn(M, cin_s, device=dev)
        if args.a_split:
            x = ops.split_activations(x)
        wt = ops.pack_conv_weight(torch.randn(cout, cin_s, kh, kw, device=dev) * 0.05)
        if args.arith:
            wt = ops.split_weights(wt)
        b = torch.randn(cout, device=dev)
        osplit = args.out_split and cout > 4
        obuf = torch.empty(M, -(-cout // 8) * 8, device=dev) if osplit else None
        for _ in range(3):
            ops.conv2d(x, wt, b, P, h, w, cout, kh, kw, act="relu", arith=args.arith, a_split=args.a_split, out_split=osplit, out=obuf,
                       tile=None if cout <= 4 else args.tile)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            ops.conv2d(x, wt, b, P, h, w, cout, kh, kw, act="relu", arith=args.arith, a_split=args.a_split, out_split=osplit, out=obuf,
                       tile=None if cout <= 4 else args.tile)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / args.reps * 1e-3
        fl = 2.0 * M * cout * cin * kh * kw
        tot_t += t * calls
        tot_f += fl * calls
        print(f"{name:24s} {t * 1e6:9.1f} us  {fl / t / 1e12:7.1f} TFLOP/s  ({fl / 1e9:6.2f} GFLOP)")
    if tot_t > 0:
        print(f"weighted per pair-batch: {tot_t * 1e3:.2f} ms, {tot_f / tot_t / 1e12:.1f} TFLOP/s "
              f"({tot_f / tot_t / 157.3e12 * 100:.1f}% of 157.3)")


if __name__ == "__main__":
    main()
