#!/usr/bin/env python3
"""Controls for tools/race_chain.py under GPU contention: which kinds of kernel lose determinism?"""
import hashlib
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd import ops  # noqa: E402

H = W = 512
g = torch.Generator().manual_seed(1)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
K = 3
Ls = [((torch.randn(2, H, W, generator=g) * 3).cuda(), torch.rand(1, H, W, generator=g).cuda() * 0.03, torch.rand(1, H, W, generator=g).cuda()) for _ in range(K)]
Rp = [((torch.randn(2, H, W, generator=g) * 3).cuda(), torch.rand(1, H, W, generator=g).cuda() * 0.03, torch.rand(1, H, W, generator=g).cuda()) for _ in range(K)]
Rs = [torch.cat([r[0].permute(1, 2, 0), r[1].permute(1, 2, 0), r[2].permute(1, 2, 0)], 2).contiguous() for r in Rp]
grid = (torch.rand(1, H, W, 2, generator=g) * 2 - 1).cuda()
img = torch.randn(1, 4, H, W, generator=g).cuda()
idx = torch.randint(0, H * W, (H * W,), generator=g).cuda()
flat = torch.randn(H * W, 4, generator=g).cuda()


def count(name, fn):
    seen = {}
    for _ in range(reps):
        out = fn()
        torch.cuda.synchronize()
        hh = hashlib.sha1()
        for t in (out if isinstance(out, (tuple, list)) else [out]):
            if t is not None:
                hh.update(t.cpu().numpy().tobytes())
        seen[hh.hexdigest()[:8]] = seen.get(hh.hexdigest()[:8], 0) + 1
    print(f"{name:46s} distinct {len(seen):4d} {sorted(seen.values(), reverse=True)[:4]}", flush=True)


count("chain_select_packed (16-byte gathers)", lambda: ops.chain_select_packed(Ls, Rs, 0.02, want_chosen=True))
count("chain_select planar (4-byte gathers)", lambda: ops.chain_select(Ls, Rp, 0.02, want_chosen=True))
count("chain (one candidate, planar)", lambda: ops.chain(Ls[0], Rp[0]))
count("select (no gathers)", lambda: ops.select([(a, b, c) for a, b, c in Rp], 0.02, want_chosen=True))
count("torch grid_sample", lambda: F.grid_sample(img, grid, align_corners=True))
count("torch index_select rows of 16 bytes", lambda: flat.index_select(0, idx))
count("torch elementwise", lambda: img * 2 + 1)
