#!/usr/bin/env python3
"""BASELINE.json configs[2] stand-in (SURVEY 8d "C3"): the TAP-Vid protocol -- 'first' and
'strided' query modes, re-initialisation per query frame, forward + backward runs, one flow cache
per sequence -- over seeded synthetic sequences with analytic ground truth (TAP-Vid-DAVIS itself is
not reachable from the build environment).  Prints one JSON line: tracked frames/s over the whole
protocol, how many flow pairs were computed vs served from the HBM cache tier, and the metrics
(with the seeded random weights the metrics only show that the pipeline is wired; a trained
checkpoint gives the paper's numbers).

    python tools/run_tapvid_synth.py [--sequences 3] [--frames 40] [--size 512] [--tracks 24]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from mft_amd import tapvid  # noqa: E402
from mft_amd.config import load_config  # noqa: E402
from mft_amd.io import FlowCache  # noqa: E402
from mft_amd.synth import SyntheticVideo  # noqa: E402


class CountingCache(FlowCache):
    hits = misses = 0

    def read(self, left_id, right_id):
        val = super().read(left_id, right_id)
        if val[0] is None:
            CountingCache.misses += 1
        else:
            CountingCache.hits += 1
        return val


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sequences", type=int, default=3)
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--tracks", type=int, default=24)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--gpu-cache-gb", type=float, default=64.0)
    a = ap.parse_args()
    conf = load_config(REPO / "configs" / "MFT_cfg.py")
    conf.flow_config.model = None
    conf.flow_config.synthetic_weights_seed = 0
    conf.flow_config.flow_iters = a.iters
    conf.keep_result_on_device = True        # point read-out happens on the device
    tracker = conf.tracker_class(conf)
    n_tracked = 0
    t_total = 0.0
    metrics = {"first": [], "strided": []}
    for s in range(a.sequences):
        vid = SyntheticVideo(a.size, a.size, n_frames=a.frames, seed=100 + s)
        occ, pts, frames = tapvid.synthetic_sequence(vid, n_tracks=a.tracks, seed=s)
        video = [np.ascontiguousarray(f) for f in frames]
        cache = CountingCache(None, max_GPU_RAM_MB=a.gpu_cache_gb * 1e3)
        for mode in ("first", "strided"):
            gt = (tapvid.sample_queries_first(occ, pts, frames) if mode == "first"
                  else tapvid.sample_queries_strided(occ, pts, frames, query_stride=5))
            q = np.round(gt["query_points"][0]).astype(np.int64)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = tapvid.run_sequence(tracker, video, q, mode, flow_cache=cache, device="cuda")
            torch.cuda.synchronize()
            t_total += time.perf_counter() - t0
            starts = np.unique(q[:, 0])
            n_tracked += sum((a.frames - int(st)) + (int(st) + 1 if mode == "strided" else 0) for st in starts)
            m = tapvid.evaluate(out, gt, mode)
            metrics[mode].append({k: float(v[0]) for k, v in m.items() if k.startswith("average") or k == "occlusion_accuracy"})
        cache.clear(clear_disk=False)
    mean = {mode: {k: float(np.mean([m[k] for m in ms])) for k in ms[0]} for mode, ms in metrics.items()}
    print(json.dumps({"workload": f"TAP-Vid protocol on {a.sequences} synthetic {a.size}x{a.size} sequences of {a.frames} frames, "
                                  f"{a.tracks} tracks, first + strided (stride 5) queries, {a.iters} RAFT iters",
                      "tracker_frames": n_tracked, "seconds": t_total, "frames_per_s": n_tracked / t_total,
                      "flow_pairs_from_cache": CountingCache.hits, "flow_pairs_computed": CountingCache.misses,
                      "metrics": mean}))


if __name__ == "__main__":
    main()
