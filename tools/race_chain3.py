#!/usr/bin/env python3
"""What do the odd results of chain_select_packed under contention look like?  K = 1; for every wrong pixel: the taps it should
have read, and whether the wrong value is explained by zeroed taps / neighbouring pixels' values / stale inputs."""
import hashlib
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd import ops  # noqa: E402

H = W = 512
g = torch.Generator().manual_seed(1)
L = ((torch.randn(2, H, W, generator=g) * 3).cuda(), torch.rand(1, H, W, generator=g).cuda() * 0.03, torch.rand(1, H, W, generator=g).cuda())
R = torch.cat([torch.randn(H, W, 2, generator=g) * 3, torch.rand(H, W, 1, generator=g) * 0.03, torch.rand(H, W, 1, generator=g)], 2).cuda().contiguous()
h_in = lambda: hashlib.sha1(b"".join(t.cpu().numpy().tobytes() for t in (*L, R))).hexdigest()[:8]   # noqa: E731
h0 = h_in()
outs = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 150):
    out = ops.chain_select_packed([L], [R], 0.02, want_chosen=True)
    torch.cuda.synchronize()
    outs.append([t.cpu().clone() for t in out])
print("inputs intact:", h_in() == h0)
keys = [hashlib.sha1(b"".join(t.numpy().tobytes() for t in o)).hexdigest() for o in outs]
major = max(set(keys), key=keys.count)
ref = outs[keys.index(major)]
print(f"{len(set(keys))} distinct of {len(keys)}; majority x{keys.count(major)}")
shown = 0
for n_run, (o, k) in enumerate(zip(outs, keys)):
    if k == major:
        continue
    d = (o[0] != ref[0]).any(0) | (o[2] != ref[2])[0] | (o[1] != ref[1])[0]
    idx = d.nonzero()
    xs = sorted(set(idx[:, 1].tolist()))
    ys = sorted(set(idx[:, 0].tolist()))
    print(f"run {n_run}: {len(idx)} wrong pixels; x mod 64 histogram of first 12: {[int(x) % 64 for x in idx[:12, 1]]}; ys {ys[:10]}")
    for (y, x) in idx[:3].tolist():
        print(f"   ({y},{x}) got flow {o[0][:, y, x].tolist()} occl {o[1][0, y, x].item():.5f} sig {o[2][0, y, x].item():.5f} | want flow {ref[0][:, y, x].tolist()} "
              f"occl {ref[1][0, y, x].item():.5f} sig {ref[2][0, y, x].item():.5f}")
        # is the wrong value the RIGHT value of some other pixel?
        for name, plane_o, plane_r in (("flow x", o[0][0], ref[0][0]), ("sigma", o[2][0], ref[2][0])):
            hit = (plane_r == plane_o[y, x]).nonzero()
            if len(hit):
                print(f"      {name}: the wrong value is the correct value of pixel {hit[0].tolist()}")
    shown += 1
    if shown >= 5:
        break
