"""Probe (2 ranks on one GPU over gloo): per-frame mode at 512 x 512 -- which unit of which frame differs from a local recompute?"""
import os, sys
from pathlib import Path
import numpy as np, torch, torch.distributed as dist
REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests"))
from mft_amd.config import Config
from mft_amd.MFT import MFT
from mft_amd.raft import RAFTWrapper
from mft_amd.synth import SyntheticVideo
from mft_amd.weights import make_weights
import mft_amd.dist as md
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank = dist.get_rank()
fc = Config(); fc.flow_iters = 12
flower = RAFTWrapper(fc, state_dict=make_weights(7))
ref_flower = RAFTWrapper(fc, state_dict=make_weights(7))
c = Config(); c.deltas = [np.inf, 1, 2, 4, 8]; c.occlusion_threshold = 0.02; c.delta_sharding = True
c.flow_config = Config(); c.flow_config.of_class = lambda cfg: flower
tr = MFT(c)
vid = SyntheticVideo(512, 512, n_frames=6, seed=9)
orig_finish = md.WindowSharder._finish_window
def finish(self, tracker, w):
    torch.cuda.synchronize()
    u = 0
    owner = {}
    for rr, (o, cnt) in enumerate(w["shares"]):
        for s_ in range(cnt): owner[o + s_] = (rr, s_)
    for j, fid in enumerate(w["frame_ids"]):
        for k, (_, left_id, _) in enumerate(w["plans"][j]):
            rr, s_ = owner[u]; u += 1
            got = w["recv"][rr, s_]
            limg = vid[left_id]
            want = ref_flower.compute_pairs([(None, limg, None, w["imgs"][j])], packed_out=True, planar=False)[0][3].clone()
            want2 = ref_flower.compute_pairs([(None, limg, None, w["imgs"][j])], packed_out=True, planar=False)[0][3].clone()
            own = tracker.flower.compute_pairs([(left_id, limg, fid, w["imgs"][j])], packed_out=True, planar=False)[0][3].clone()
            torch.cuda.synchronize()
            print(f"[rank {rank}] frame {fid} unit {k} ({left_id}->{fid}) by rank {rr} slot {s_}: got-vs-ref {float((got - want).abs().max()):.3e} "
                  f"ref-vs-ref {float((want2 - want).abs().max()):.3e} own-vs-ref {float((own - want).abs().max()):.3e} got-vs-own {float((got - own).abs().max()):.3e}", flush=True)
    return orig_finish(self, tracker, w)
md.WindowSharder._finish_window = finish
tr.init(vid[0])
for i in range(1, 5):
    tr.track(vid[i])
dist.barrier(); dist.destroy_process_group()
