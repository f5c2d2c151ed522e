#!/usr/bin/env python3
"""Which intermediate of a refinement differs between repeated identical calls under GPU contention?  Run several copies at once.
After every call the workspace is hashed region by region (mftx_raft_workspace_layout_for); regions with more than one distinct
hash over the repetitions are reported in workspace order."""
import argparse
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd import _lib  # noqa: E402
from mft_amd.config import AttrDict, Config  # noqa: E402
from mft_amd.raft import RAFTWrapper  # noqa: E402
from mft_amd.synth import SyntheticVideo  # noqa: E402
from mft_amd.weights import make_weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=100)
    ap.add_argument("--iters", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=1)
    ap.add_argument("--opt", nargs="*", default=[])
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    opts = {k: int(v) for k, v in (kv.split("=") for kv in a.opt)}
    c = Config()
    c.flow_iters = a.iters
    c.raft_params = AttrDict(engine_options=opts)
    fl = RAFTWrapper(c, state_dict=make_weights(7))
    vid = SyntheticVideo(512, 512, n_frames=9, seed=9)
    pairs = [(i, vid[i], 8, vid[8]) for i in range(a.pairs)]
    fl.compute_pairs(pairs, packed_out=True, planar=False)
    eng = fl.engine
    P, h, w = a.pairs, 64, 64
    offs = (C.c_size_t * 19)()
    _lib.check(_lib.load().mftx_raft_workspace_layout_for(eng._h, P, h, w, offs, 19), "layout")
    order = sorted(range(19), key=lambda i: offs[i])
    ends = {order[k]: (offs[order[k + 1]] if k + 1 < 19 else eng._ws.numel()) for k in range(19)}
    seen = {name: {} for name in eng.REGIONS}
    seen["OUT"] = {}
    words = eng._ws.view(torch.int32)
    for _ in range(a.reps):
        out = fl.compute_pairs(pairs, packed_out=True, planar=False)
        # checksums on the device (integer sums: order-independent), one small download per call -- the GPU stays contended
        sums = torch.stack([words[offs[i] // 4: ends[i] // 4].to(torch.int64).sum() for i in range(19)] +
                           [torch.stack([o[3] for o in out]).view(torch.int32).to(torch.int64).sum()]).cpu().tolist()
        for i, name in enumerate(eng.REGIONS):
            seen[name][sums[i]] = seen[name].get(sums[i], 0) + 1
        seen["OUT"][sums[19]] = seen["OUT"].get(sums[19], 0) + 1
    bad = [(eng.REGIONS[i], len(seen[eng.REGIONS[i]])) for i in order if len(seen[eng.REGIONS[i]]) > 1]
    print(f"{a.tag} iters={a.iters} P={P} opts={opts}: OUT distinct {len(seen['OUT'])}; regions with > 1 hash (workspace order): {bad}", flush=True)


if __name__ == "__main__":
    main()
