#!/usr/bin/env python3
"""What the z|r gate GEMM costs beyond a plain layer of its shape (7 pairs, split arithmetic, split-form A): one
contiguous 256-channel input -> the engine's two segments out of the 384-wide hx rows -> + the hoisted addend ->
+ the GRU epilogue's traffic is in the engine itself (tools/bench_pairs.py + rocprofv3)."""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd import _lib, ops  # noqa: E402

lib = _lib.load()
P, h, w = 7, 64, 64
M = P * h * w
dev = "cuda"


def run(a0, lda0, c0, a1, lda1, c1, wt, bias, addend, out, N, kh, kw, out_split=0):
    d = _lib.ConvDesc()
    d.a0, d.lda0, d.c0 = a0, lda0, c0
    d.a1, d.lda1, d.c1 = a1, lda1, c1
    d.wpk, d.bias = wt.data_ptr(), (bias.data_ptr() if bias is not None else None)
    d.out, d.ldo = out.data_ptr(), out.shape[1]
    d.P, d.h, d.w, d.N, d.kh, d.kw = P, h, w, N, kh, kw
    d.act, d.out_scale = 1, 1.0
    d.addend, d.ld_addend = (addend.data_ptr(), addend.shape[1]) if addend is not None else (None, 0)
    d.arith, d.a_split, d.out_split = 1, 1, out_split

    def f():
        _lib.check(lib.mftx_conv2d(C.byref(d), torch.cuda.current_stream().cuda_stream), "conv2d")
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


for name, N, kh, kw in (("z|r 1x5 256->256", 256, 1, 5), ("q 1x5 256->128", 128, 1, 5)):
    wt = ops.split_weights(ops.pack_conv_weight(torch.randn(N, 256, kh, kw, device=dev) * 0.05))
    b = torch.randn(N, device=dev)
    x = ops.split_activations(torch.randn(M, 256, device=dev))
    hx = ops.split_activations(torch.randn(M, 384, device=dev))
    add = torch.randn(M, N, device=dev)
    out = torch.empty(M, N, device=dev)
    e = hx.element_size()
    print(f"{name}: one segment of 256 channels out of hx (row stride 384) {run(hx.data_ptr(), 384, 256, None, 0, 0, wt, b, None, out, N, kh, kw):.1f} us | "
          f"two segments out of the contiguous input (row stride 256) {run(x.data_ptr(), 256, 128, x.data_ptr() + 128 * e, 256, 128, wt, b, None, out, N, kh, kw):.1f}")
    print(f"{name}: one contiguous input {run(x.data_ptr(), 256, 256, None, 0, 0, wt, b, None, out, N, kh, kw):.1f} us | "
          f"two segments of hx {run(hx.data_ptr(), 384, 128, hx.data_ptr() + 256 * e, 384, 128, wt, b, None, out, N, kh, kw):.1f} | "
          f"+ addend {run(hx.data_ptr(), 384, 128, hx.data_ptr() + 256 * e, 384, 128, wt, None, add, out, N, kh, kw):.1f} | "
          f"+ split output {run(hx.data_ptr(), 384, 128, hx.data_ptr() + 256 * e, 384, 128, wt, None, add, out, N, kh, kw, 1):.1f}")
