#!/usr/bin/env python3
"""TAP-Vid run + evaluation with the reference's command line (MFT/runners/run_MFT_tapvid.py:29-47,
eval_MFT_tapvid.py): tracklet pickles under --export/<tracker>/results, metrics under .../eval.

    MFT_TAPVID_DAVIS=/data/tapvid_davis.pkl python tools/run_MFT_tapvid.py \\
        dataset_configs/pkl-tapvid-davis-256x256_512x512.py configs/MFT_cfg.py --export ./export --cache ./cache
    python tools/run_MFT_tapvid.py --synthetic 3 ...      # a TAP-Vid-shaped pickle of seeded synthetic sequences instead

Everything it does lives in mft_amd.tapvid (run_dataset / evaluate_dataset)."""
import argparse
import json
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))


def main():
    ap = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    ap.add_argument("dataset", type=Path, nargs="?", default=REPO / "dataset_configs" / "pkl-tapvid-davis-256x256_512x512.py",
                    help="dataset config")
    ap.add_argument("trackers", type=Path, nargs="*", default=[REPO / "configs" / "MFT_cfg.py"],
                    help="tracker configs, all sharing one flow_config")
    ap.add_argument("--export", type=Path, default=Path("./export"))
    ap.add_argument("--cache", type=Path, default=Path("./cache"))
    ap.add_argument("-c", "--cont", action="store_true", help="skip already computed sequences")
    ap.add_argument("--debug", action="store_true")
    ap.add_argument("--mode", choices=["first", "strided", "both"], default="both")
    ap.add_argument("--write_flow", action="store_true", help="write flowou for the frame 0 template")
    ap.add_argument("-rcl", "--ram_cache_limit", type=int, default=30, help="RAM cache limit in GB")
    ap.add_argument("-gcl", "--gpu_cache_limit", type=int, default=64, help="HBM cache limit in GB (reference default: 5)")
    ap.add_argument("--seq", nargs="+", help="sequence subset")
    ap.add_argument("--no-eval", action="store_true")
    ap.add_argument("--synthetic", type=int, default=0, metavar="N",
                    help="no dataset: N seeded synthetic 256 x 256 sequences written as a TAP-Vid pickle, seeded synthetic weights")
    ap.add_argument("--synthetic-frames", type=int, default=40)
    a = ap.parse_args()
    from mft_amd import tapvid
    from mft_amd.config import load_config
    dconf = load_config(a.dataset)
    configs = [load_config(p) for p in a.trackers]
    tmp = None
    if a.synthetic:
        from mft_amd.synth import SyntheticVideo
        tmp = tempfile.TemporaryDirectory()
        path = Path(tmp.name) / "synthetic_tapvid.pkl"
        tapvid.synthetic_pickle(path, {f"synth-{i:02d}": SyntheticVideo(256, 256, n_frames=a.synthetic_frames, seed=100 + i)
                                       for i in range(a.synthetic)})
        dconf.pickles = [path]
        for c in configs:
            c.flow_config.model = None
            c.flow_config.synthetic_weights_seed = 0
    for c in configs:
        c.keep_result_on_device = True          # the point read-out happens on the device
    # the reference compares configs with ==; sharing ONE flow config object keeps that check meaningful after the edits above
    for c in configs[1:]:
        assert c.flow_config == configs[0].flow_config
    t0 = time.perf_counter()
    done = tapvid.run_dataset(dconf, configs, a.export, a.cache, mode=a.mode, cont=a.cont, seqs=a.seq, write_flow=a.write_flow,
                              ram_cache_limit=a.ram_cache_limit, gpu_cache_limit=a.gpu_cache_limit, debug=a.debug)
    dt = time.perf_counter() - t0
    summary = {"dataset": str(dconf.name), "scaling": dconf.scaling, "results": len(done),
               "skipped": sum(d["skipped"] for d in done), "seconds": dt}
    if not a.no_eval:
        m = tapvid.evaluate_dataset(dconf, configs, a.export, mode=a.mode)
        summary["metrics"] = {mode: {name: {k: float(np.mean([r[k] for r in rows])) for k in
                                            ("average_jaccard", "average_pts_within_thresh", "occlusion_accuracy")}
                                     for name, rows in per.items() if rows} for mode, per in m.items()}
    print(json.dumps(summary))
    if tmp is not None:
        tmp.cleanup()


if __name__ == "__main__":
    main()
