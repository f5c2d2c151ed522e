#!/usr/bin/env python3
"""Soak run: 252 tracked frames (4 passes over a 64-frame synthetic video) with raft_params.check_finite -- every result is
checked for non-finite values (which is what an operand leaving the split arithmetic's range would produce), graphs
replaying, encoders on their side stream.  Prints the frame count, the (synchronising) rate and the graph statistics.

    python tools/soak.py"""
import sys, time, torch
sys.path.insert(0, '.')
from pathlib import Path
from mft_amd.config import load_config
from mft_amd.synth import SyntheticVideo
conf = load_config(Path('configs/MFT_cfg.py'))
conf.flow_config.model = None
conf.flow_config.synthetic_weights_seed = 0
conf.flow_config.raft_params.check_finite = True
conf.flow_config.async_encode = True
conf.keep_result_on_device = True
tr = conf.tracker_class(conf)
vid = SyntheticVideo(512, 512, n_frames=64, seed=3)
tr.init(vid[0])
t0 = time.time()
n = 0
for rep in range(4):
    for i in range(1, 64):
        m = tr.track(vid[i]); n += 1
torch.cuda.synchronize()
r = m.result
print('frames', n, 'fps', round(n / (time.time() - t0), 1), 'finite', bool(torch.isfinite(r.flow).all()), 'graphs', tr.flower.engine.graph_stats())
