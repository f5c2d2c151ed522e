#!/usr/bin/env python3
"""Determinism stress of the fused lookup + convc1 kernel: the same launch repeated, with and without a cache flush in
between; every result must equal the first one bit for bit."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd import ops  # noqa: E402

dev = "cuda"
flush = torch.empty(64 << 20, dtype=torch.float32, device=dev)
for P, h, w in ((2, 33, 50), (7, 64, 64), (3, 17, 23), (1, 64, 64), (5, 46, 62)):
    g = torch.Generator().manual_seed(P * 1000 + h)
    N = h * w
    f1 = torch.randn(P, N, 256, generator=g).to(dev)
    f2 = torch.randn(P, N, 256, generator=g).to(dev)
    lv = ops.corr_pyramid(f1, f2, h, w, arith=ops.ARITH_SPLIT)
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    coords = (torch.stack([xs, ys], -1).reshape(1, N, 2).float() + 6 * torch.randn(P, N, 2, generator=g)).to(dev).contiguous()
    wpk = ops.pack_conv_weight((torch.randn(256, 324, 1, 1, generator=g) * 0.05).to(dev))
    bias = torch.randn(256, generator=g).to(dev)
    wf = ops.pack_lookup_convc1_weights(wpk)
    ref = ops.corr_lookup_convc1(lv, coords, h, w, wf, bias).clone()
    feats = ops.corr_lookup(lv, coords, h, w).reshape(P * N, 324)
    apart = ops.conv2d(feats, ops.split_weights(wpk), bias, P, h, w, 256, 1, 1, act="relu", arith=ops.ARITH_SPLIT)
    err = float((ref - apart).abs().max()) / float(apart.abs().max())
    print(f"P={P} {h}x{w}: fused vs lookup -> convc1: max rel diff {err:.2e}, nan {bool(torch.isnan(ref).any())}")
    assert err < 1e-5
    bad = 0
    for it in range(120):
        if it % 3 == 0:
            flush.fill_(float(it))
        if it % 7 == 0:
            torch.cuda.synchronize()
        out = ops.corr_lookup_convc1(lv, coords, h, w, wf, bias)
        if not torch.equal(out, ref):
            d = (out != ref)
            rows = d.any(1).nonzero().flatten()
            bad += 1
            if bad <= 3:
                print(f"  P={P} {h}x{w} it {it}: {int(d.sum())} values differ in {len(rows)} rows, first rows {rows[:6].tolist()}, "
                      f"cols {d[rows[0]].nonzero().flatten()[:8].tolist()}, nan {bool(torch.isnan(out).any())}, "
                      f"max diff {float((out - ref).abs().max()):.3e}")
    print(f"P={P} {h}x{w}: {bad} of 120 launches differ from the first")
