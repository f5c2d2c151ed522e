#!/usr/bin/env python3
"""Repeat-launch, on fixed inputs at the benchmark's size (7 pairs of 64 x 64 cells, 512 x 512 frames), the kernels outside chain.hip that
carry EXEC-masked gathers -- the stand-alone correlation lookup, the on-demand lookup, the convex upsampler, both encoders and
the cache codec -- and count distinct results.  Run beside load generators (tools/race_kernels.py --load-seconds N): the chain race of
round 5 (profiles/r5q_chain_race.txt) showed only then.  Their translation units are built without packed-fp32 instructions since
round 6 (csrc/Makefile: NOPK); this is the behavioural half of that guard.

    python tools/race_masked.py [reps]"""
import hashlib
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd import ops  # noqa: E402
from mft_amd.synth import SyntheticVideo  # noqa: E402
from mft_amd.weights import make_weights  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
P, h, w = 7, 64, 64
N = h * w
dev = torch.device("cuda")
g = torch.Generator().manual_seed(5)


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(dev)


f1, f2 = rnd(P, N, 256, scale=0.5), rnd(P, N, 256, scale=0.5)
lv = ops.corr_pyramid(f1, f2, h, w, arith=ops.ARITH_SPLIT)
ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
coords = (torch.stack([xs, ys], -1).reshape(1, N, 2) + torch.randn(P, N, 2, generator=g) * 6).to(dev).contiguous()
f2_levels = ops.fmap_pyramid(f2, h, w)
flow_lr, ou, mask = rnd(P * N, 2, scale=2.0), rnd(P * N, 4), rnd(P * N, 576)
sd = {k: torch.from_numpy(v) for k, v in make_weights(0).items()}
fnet = ops.EncoderEngine(sd, "fnet", True, dev)
cnet = ops.EncoderEngine(sd, "cnet", False, dev)
img = torch.from_numpy(SyntheticVideo(512, 512, n_frames=2, seed=9)[1]).to(dev)
plane = rnd(512, 512, scale=3.0)


def count(name, fn, n=reps):
    seen = {}
    for _ in range(n):
        out = fn()
        torch.cuda.synchronize()
        hh = hashlib.sha1()
        for t in (out if isinstance(out, (tuple, list)) else [out]):
            if t is not None:
                hh.update(t.cpu().numpy().tobytes())
        k = hh.hexdigest()[:8]
        seen[k] = seen.get(k, 0) + 1
    print(f"{name:46s} distinct {len(seen):4d} {sorted(seen.values(), reverse=True)[:4]}", flush=True)
    return len(seen)


def codec():
    q, lohi = ops.quantize_u16(plane)
    lo, hi = lohi.tolist()
    return q.view(torch.int16), ops.dequantize_u16(q, lo, hi)


bad = 0
bad += count("corr_lookup (stand-alone, 7 x 4096 cells)", lambda: ops.corr_lookup(lv, coords, h, w)) != 1
bad += count("corr_lookup_ondemand (7 x 4096 cells)", lambda: ops.corr_lookup_ondemand(f1, f2_levels, coords, h, w), max(50, reps // 10)) != 1
bad += count("convex_upsample (7 x 512 x 512, packed too)", lambda: ops.convex_upsample(flow_lr, ou, mask, P, h, w, want_packed=True)) != 1
bad += count("encoder fnet (512 x 512)", lambda: fnet.forward(img)) != 1
bad += count("encoder cnet (512 x 512)", lambda: cnet.forward(img)) != 1
bad += count("codec quantize + dequantize (512 x 512)", codec) != 1
sys.exit(1 if bad else 0)
