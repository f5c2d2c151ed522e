#!/usr/bin/env python3
"""Frames/s of the SHIPPED configuration (configs/MFT_cfg.py, nothing set but the stand-in weights and device-resident results): host
frames (numpy) in, no per-frame synchronisation.  What a user of the reference's demo loop gets without touching a switch.

    python tools/default_config_rate.py [frames]        (MFTX_FRAMES_IN_FLIGHT=1 for the A/B)"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd.config import load_config  # noqa: E402
from mft_amd.synth import SyntheticVideo  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
conf = load_config(Path(__file__).resolve().parents[1] / "configs" / "MFT_cfg.py")
conf.flow_config.model = None
conf.flow_config.synthetic_weights_seed = 0
conf.keep_result_on_device = True
tr = conf.tracker_class(conf)
vid = SyntheticVideo(512, 512, n_frames=64, seed=3)
frames = [vid[i % 64] for i in range(n + 41)]
tr.init(frames[0])
for i in range(1, 41):
    tr.track(frames[i])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(41, 41 + n):
    tr.track(frames[i])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"shipped config, host frames, {n} frames: {n / dt:.1f} frames/s (frames in flight {tr.flower._fif}, encode stream {'on' if tr.flower._enc_stream is not None else 'off'})")
