"""Probe: does a pair's result depend on the batch it rides in (512 x 512, 12 iterations)?"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from mft_amd.config import Config
from mft_amd.raft import RAFTWrapper
from mft_amd.synth import SyntheticVideo
from mft_amd.weights import make_weights
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 12
c = Config(); c.flow_iters = iters
fl = RAFTWrapper(c, state_dict=make_weights(7))
vid = SyntheticVideo(512, 512, n_frames=9, seed=9)
lefts = [(i, vid[i]) for i in range(7)]
ref = fl.compute_pairs([(k, im, 8, vid[8]) for k, im in lefts], packed_out=True, planar=False)
for P in range(1, 8):
    for off in (0, 7 - P):
        got = fl.compute_pairs([(k, im, 8, vid[8]) for k, im in lefts[off:off + P]], packed_out=True, planar=False)
        bad = [off + i for i in range(P) if not torch.equal(got[i][3], ref[off + i][3])]
        d = max(float((got[i][3] - ref[off + i][3]).abs().max()) for i in range(P))
        print(f"P={P} off={off}: differing pairs {bad} max abs diff {d:.3e}", flush=True)
