#!/usr/bin/env python3
"""flow_config.frames_in_flight under GPU contention: a short tracker sequence, repeated, with one lane and with two -- every repeat
of either must give the one result (a hash over all frames' flow / occlusion / sigma).  Run beside load generators
(tools/race_kernels.py --load-seconds N), or as several concurrent copies of itself.

    python tools/race_lanes.py --reps 20 [--size 512] [--frames 6] [--iters 12] [--sync] [--numpy] [--no-async-encode]
"""
import argparse
import hashlib
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd.config import Config  # noqa: E402
from mft_amd.MFT import MFT  # noqa: E402
from mft_amd.raft import RAFTWrapper  # noqa: E402
from mft_amd.synth import SyntheticVideo  # noqa: E402
from mft_amd.weights import make_weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--lanes", type=int, nargs="*", default=[1, 2])
    ap.add_argument("--sync", action="store_true", help="results to the host after every frame (keep_result_on_device = False)")
    ap.add_argument("--numpy", action="store_true", help="host frames (uploaded by the plugin) instead of device tensors")
    ap.add_argument("--no-async-encode", action="store_true")
    ap.add_argument("--tag", default="")
    ap.add_argument("--stages", action="store_true", help="also hash every frame's encodings and every flow batch (synchronises per call)")
    a = ap.parse_args()
    vid = SyntheticVideo(a.size, a.size, n_frames=a.frames, seed=9)
    sd = make_weights(7)
    seen = {}
    for lanes in a.lanes:
        fc = Config()
        fc.flow_iters = a.iters
        fc.async_encode = not a.no_async_encode
        fc.frames_in_flight = lanes
        fl = RAFTWrapper(fc, state_dict=sd)
        c = Config()
        c.deltas = [np.inf, 1, 2, 4, 8, 16, 32]
        c.occlusion_threshold = 0.02
        c.keep_result_on_device = not a.sync
        c.flow_config = Config()
        c.flow_config.of_class = lambda cfg, fl=fl: fl
        tr = MFT(c)
        hashes = {}
        stage = {"enc": {}, "flow": {}}          # --stages: per (frame / call) distinct hashes of the encodings and of the flow batches
        if a.stages:
            orig = fl.compute_pairs
            calls = [0]

            def wrapped(pairs, *args, _orig=orig, **kw):
                out = _orig(pairs, *args, **kw)
                torch.cuda.synchronize()
                k = calls[0] % (a.frames - 1)
                calls[0] += 1
                hh = hashlib.sha1()
                for o in out:
                    hh.update(o[3].cpu().numpy().tobytes())
                stage["flow"].setdefault(k, set()).add(hh.hexdigest()[:8])
                for key, f in sorted(fl._frames.items()):
                    he = hashlib.sha1(torch.cat([f.fmap.reshape(-1), f.net.reshape(-1), f.inp.reshape(-1)]).cpu().numpy().tobytes()).hexdigest()[:8]
                    stage["enc"].setdefault(key, set()).add(he)
                return out
            fl.compute_pairs = wrapped
        for rep in range(a.reps):
            frames = [vid[i] if a.numpy else torch.from_numpy(vid[i]).cuda() for i in range(a.frames)]
            tr.init(frames[0])
            res = [tr.track(f).result for f in frames[1:]]
            torch.cuda.synchronize()
            per_frame = []
            for r in res:
                hh = hashlib.sha256()
                for t in (r.flow, r.occlusion, r.sigma):
                    hh.update(t.cpu().numpy().tobytes())
                per_frame.append(hh.hexdigest()[:10])
            key = ",".join(per_frame)
            hashes[key] = hashes.get(key, 0) + 1
        seen[lanes] = hashes
        print(f"{a.tag} lanes={lanes} distinct {len(hashes)} of {a.reps} reps: " + " | ".join(f"{k} x{v}" for k, v in list(hashes.items())[:4]), flush=True)
        if a.stages:
            print(f"{a.tag}   distinct encodings per frame: " + " ".join(f"{k}:{len(v)}" for k, v in sorted(stage["enc"].items())))
            print(f"{a.tag}   distinct flow batches per frame: " + " ".join(f"{k + 1}:{len(v)}" for k, v in sorted(stage["flow"].items())))
    keys = {k for h in seen.values() for k in h}
    print(f"{a.tag} overall distinct {len(keys)}")
    return 0 if len(keys) == 1 else 1


if __name__ == "__main__":
    sys.exit(main())
