"""Worker for the single-node RCCL test of the delta-sharded tracker (GPU)."""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))

from mft_amd.config import Config  # noqa: E402
from mft_amd.MFT import MFT  # noqa: E402
from mft_amd.raft import RAFTWrapper  # noqa: E402
from mft_amd.synth import SyntheticVideo  # noqa: E402
from mft_amd.weights import make_weights  # noqa: E402


def run(flower, sharding, n_frames):
    c = Config()
    c.deltas = [np.inf, 1, 2, 4, 8]
    c.occlusion_threshold = 0.02
    c.delta_sharding = sharding
    c.flow_config = Config()
    c.flow_config.of_class = lambda cfg: flower
    tr = MFT(c)
    vid = SyntheticVideo(128, 160, n_frames=n_frames, seed=4)
    tr.init(vid[0])
    out = {}
    for i in range(1, n_frames):
        res = tr.track(vid[i]).result
        out[f"flow{i}"], out[f"occl{i}"], out[f"sigma{i}"] = res.flow.numpy(), res.occlusion.numpy(), res.sigma.numpy()
        out[f"chosen{i}"] = tr.last_chosen.cpu().numpy()
    return out


if __name__ == "__main__":
    outdir = Path(sys.argv[1])
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    fc = Config()
    fc.flow_iters = 4
    flower = RAFTWrapper(fc, state_dict=make_weights(7))
    rank, world = dist.get_rank(), dist.get_world_size()
    res = run(flower, "force" if world == 1 else True, 10)
    np.savez(outdir / f"rank{rank}.npz", **res)
    if rank == 0:
        np.savez(outdir / "single.npz", **run(flower, False, 10))
    dist.barrier()
    dist.destroy_process_group()
