#!/usr/bin/env python3
"""Determinism under GPU contention: run this as TWO (or more) concurrent processes on one GPU.  Each repeats the same 512 x 512 flow
pair under a list of engine-option sets and reports how many DISTINCT results it saw per set (1 = deterministic).  A timing-dependent
race inside the engine that never shows with the GPU to itself shows here, because a second process's kernels perturb the order in
which workgroups and concurrent kernels run.

    for i in 0 1; do python tools/race_probe.py --reps 8 & done; wait
"""
import argparse
import hashlib
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd.config import AttrDict, Config  # noqa: E402
from mft_amd.raft import RAFTWrapper  # noqa: E402
from mft_amd.synth import SyntheticVideo  # noqa: E402
from mft_amd.weights import make_weights  # noqa: E402

SETS = {
    "default": {},
    "graph=0": {"graph": 0},
    "fork=0": {"fork": 0},
    "fuse_gru=0": {"fuse_gru": 0},
    "fuse_flow=0": {"fuse_flow": 0},
    "fuse_head=0": {"fuse_head": 0},
    "fuse_lookup=0": {"fuse_lookup": 0},
    "fuse_ou=0": {"fuse_ou": 0},
    "tile_conv=0": {"tile_conv": 0},
    "tile_volume=0": {"tile_volume": 0},
    "fuse_head=0,fuse_gru=0": {"fuse_head": 0, "fuse_gru": 0},
    "fuse_head=0,fuse_ou=0": {"fuse_head": 0, "fuse_ou": 0},
    "fuse_head=2": {"fuse_head": 2},
    "tile_cells=128": {"tile_cells": 128},
    "tile_cells=64": {"tile_cells": 64},
    "tile_cells=32": {"tile_cells": 32},
    "all_off": {"graph": 0, "fork": 0, "fuse_gru": 0, "fuse_flow": 0, "fuse_head": 0, "fuse_lookup": 0, "fuse_ou": 0, "tile_conv": 0, "tile_volume": 0},
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--pairs", type=int, default=1)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--sets", nargs="*", default=list(SETS))
    ap.add_argument("--arith", default="split")
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    vid = SyntheticVideo(a.size, a.size, n_frames=9, seed=9)
    sd = make_weights(7)
    for name in a.sets:
        c = Config()
        c.flow_iters = a.iters
        c.raft_params = AttrDict(engine_options=dict(SETS[name]), arith=a.arith)
        fl = RAFTWrapper(c, state_dict=sd)
        pairs = [(i, vid[i], 8, vid[8]) for i in range(a.pairs)]
        seen, enc = {}, {}
        for _ in range(a.reps):
            fl.reset_cache()
            out = fl.compute_pairs(pairs, packed_out=True, planar=False)
            torch.cuda.synchronize()
            h = hashlib.sha1(torch.stack([o[3] for o in out]).cpu().numpy().tobytes()).hexdigest()[:10]
            seen[h] = seen.get(h, 0) + 1
            f = fl._frames[8]
            he = hashlib.sha1(torch.cat([f.fmap.reshape(-1), f.net.reshape(-1), f.inp.reshape(-1)]).cpu().numpy().tobytes()).hexdigest()[:10]
            enc[he] = enc.get(he, 0) + 1
        print(f"{a.tag} {name:14s} pairs={a.pairs} distinct results {len(seen)} {sorted(seen.values(), reverse=True)}  distinct encodings {len(enc)}", flush=True)


if __name__ == "__main__":
    main()
