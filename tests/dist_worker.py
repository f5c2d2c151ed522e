"""Worker for the 2-rank gloo test of the delta-sharded tracker (CPU)."""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))

import golden_inputs as gi  # noqa: E402
from test_host_logic import StubFlower, make_tracker  # noqa: E402


def run(sharded, n_frames=12):
    tr = make_tracker(StubFlower(), delta_sharding=sharded)
    tr.init(gi.id_image(0))
    out = {}
    for i in range(1, n_frames):
        res = tr.track(gi.id_image(i)).result
        out[f"flow{i}"] = res.flow.numpy()
        out[f"occl{i}"] = res.occlusion.numpy()
        out[f"sigma{i}"] = res.sigma.numpy()
        out[f"chosen{i}"] = tr.last_chosen.numpy()
    return out


if __name__ == "__main__":
    outdir = Path(sys.argv[1])
    torch.set_num_threads(2)
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    res = run(sharded=True)
    np.savez(outdir / f"rank{rank}.npz", **res)
    if rank == 0:
        np.savez(outdir / "single.npz", **run(sharded=False))
    dist.barrier()
    dist.destroy_process_group()
