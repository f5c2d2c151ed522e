#!/usr/bin/env python3
"""Track every pixel of a video with MFT and write the point / edit overlays -- the reference's ``demo.py`` on
the MI355X engine.

    python demo.py --video <dir of PNG frames | frames.npy> [--edit edit.png] [--out demo_out/] [--synthetic]

Differences forced by the environment: no OpenCV here, so the input is a directory of PNG frames or a ``.npy``
frame array (video containers work when cv2 is importable) and the overlays are written as numbered PNGs instead of
an mp4; ``--synthetic`` tracks the seeded synthetic video with seeded stand-in weights (no checkpoint ships with
this build).  Frames go to the GPU through a pinned upload ring (``mft_amd/video.py:FrameRing``).
"""
import argparse
import logging
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

from mft_amd.config import load_config  # noqa: E402
from mft_amd.point_tracking import convert_to_point_tracking  # noqa: E402
from mft_amd import video as vio  # noqa: E402
from mft_amd import vis  # noqa: E402

logger = logging.getLogger("demo")


def parse_arguments():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('-v', '--verbose', action='store_true')
    ap.add_argument('--video', type=Path, default=Path('demo_in/frames'), help='directory of PNG frames or .npy / .npz frame array')
    ap.add_argument('--edit', type=Path, default=Path('demo_in/edit.png'), help='RGBA png with a first-frame edit')
    ap.add_argument('--config', type=Path, default=REPO / 'configs' / 'MFT_cfg.py')
    ap.add_argument('--out', type=Path, default=Path('demo_out/'))
    ap.add_argument('--grid_spacing', type=int, default=30)
    ap.add_argument('--synthetic', type=int, default=0, metavar='N', help='track N frames of the seeded synthetic video')
    ap.add_argument('--synthetic_weights_seed', type=int, default=None, help='run on seeded stand-in weights')
    return ap.parse_args()


def run(args):
    logging.basicConfig(level=logging.DEBUG if args.verbose else logging.INFO,
                        format="[%(asctime)s] %(levelname)s:%(name)s:%(message)s")
    config = load_config(args.config)
    if args.synthetic or args.synthetic_weights_seed is not None:
        config.flow_config.model = None
        config.flow_config.synthetic_weights_seed = args.synthetic_weights_seed or 0
    config.keep_result_on_device = True
    tracker = config.tracker_class(config)
    if args.synthetic:
        from mft_amd.synth import SyntheticVideo
        src = SyntheticVideo(512, 512, n_frames=args.synthetic, seed=0)
        frames = [src[i] for i in range(args.synthetic)]
        name = "synthetic"
    else:
        frames = list(vio.get_video_frames(args.video))
        name = args.video.stem
    logger.info("tracking %d frames", len(frames))
    from mft_amd.results import FlowOUTrackingResult
    results, host_results, queries = [], [], None
    drain = vio.ResultDrain()
    import torch
    up = [torch.cuda.current_stream()]                   # the streams frames are uploaded on (FrameRing's reuse guard)
    if getattr(tracker.flower, "_enc_stream", None) is not None:
        up.append(tracker.flower._enc_stream)
    for i, dev_frame in enumerate(vio.FrameRing(frames, streams=up)):
        if i == 0:
            meta = tracker.init(dev_frame)
            meta.result = meta.result.cuda()
            queries = vis.get_queries(frames[0].shape[:2], args.grid_spacing).cuda()
        else:
            meta = tracker.track(dev_frame)
        coords, occlusions = convert_to_point_tracking(meta.result, queries)
        drain.submit(meta.result)
        host_results.append(FlowOUTrackingResult(*drain.collect(copy=True), validate=False))
        results.append((coords, occlusions))
    edit = vio.imread_unchanged(args.edit) if args.edit.exists() else None
    for i, frame in enumerate(frames):
        coords, occlusions = results[i]
        result = host_results[i]
        vio.imwrite_bgr(args.out / f"{name}_points" / f"{i:05d}.png", vis.draw_dots(frame, coords, occlusions))
        if edit is not None:
            vio.imwrite_bgr(args.out / f"{name}_edit" / f"{i:05d}.png", vis.draw_edit(frame, result, edit))
    logger.info("wrote %s", args.out)
    return 0


if __name__ == '__main__':
    sys.exit(run(parse_arguments()))
