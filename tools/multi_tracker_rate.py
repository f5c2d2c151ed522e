#!/usr/bin/env python3
"""Does the N-th tracker of one process run as fast as the first?  (HIP hands hardware queues to streams as they are created; the
plugin's streams are shared per process so that every instance keeps the first one's queues -- mft_amd/raft.py: _shared_stream.)

    python tools/multi_tracker_rate.py [trackers]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd.config import load_config  # noqa: E402
from mft_amd.synth import SyntheticVideo  # noqa: E402

vid = SyntheticVideo(512, 512, n_frames=64, seed=3)
frames = [torch.from_numpy(vid[i]).cuda() for i in range(64)]
for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    conf = load_config(Path(__file__).resolve().parents[1] / "configs" / "MFT_cfg.py")
    conf.flow_config.model = None
    conf.flow_config.synthetic_weights_seed = 0
    conf.flow_config.async_encode = True            # (device frames, complete before they are passed in)
    conf.keep_result_on_device = True
    tr = conf.tracker_class(conf)
    tr.init(frames[0])
    for i in range(1, 41):
        tr.track(frames[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(120):
        tr.track(frames[1 + (40 + k) % 63])
    torch.cuda.synchronize()
    print(f"tracker {t} of this process: {120 / (time.perf_counter() - t0):.1f} frames/s", flush=True)
