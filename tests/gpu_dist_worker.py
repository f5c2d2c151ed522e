"""Worker for the single-node RCCL test of the window-sharded tracker (GPU)."""
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

N_FRAMES = 14


def run(flower, sharding, window, prefetch=False, defer=False):
    c = Config()
    c.deltas = [np.inf, 1, 2, 4, 8]
    c.occlusion_threshold = 0.02
    c.delta_sharding = sharding
    c.flow_config = Config()
    c.flow_config.of_class = lambda cfg: flower
    tr = MFT(c)
    vid = SyntheticVideo(128, 160, n_frames=N_FRAMES, seed=4)
    tr.init(vid[0])
    out, i, done = {}, 1, 1

    def record(metas):
        nonlocal done
        for m in metas:
            res = m.result
            out[f"flow{done}"], out[f"occl{done}"], out[f"sigma{done}"] = \
                res.flow.numpy(), res.occlusion.numpy(), res.sigma.numpy()
            done += 1

    while i < N_FRAMES:
        imgs = [vid[k] for k in range(i, min(i + window, N_FRAMES))]
        nxt = [vid[k] for k in range(i + window, min(i + 2 * window, N_FRAMES))] if prefetch else None
        record(tr.track_window(imgs, next_imgs=nxt, defer=defer) if window > 1 else [tr.track(imgs[0])])
        i += len(imgs)
    if defer:
        record(tr.flush_window())
    assert done == N_FRAMES
    out["final_chosen"] = tr.last_chosen.cpu().numpy()
    out["final_keys"] = np.array(sorted(tr.memory.keys()))
    return out, tr


if __name__ == "__main__":
    outdir = Path(sys.argv[1])
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    fc = Config()
    fc.flow_iters = 4
    flower = RAFTWrapper(fc, state_dict=make_weights(7))
    rank, world = dist.get_rank(), dist.get_world_size()
    sharding = "force" if world == 1 else True
    for mode, window in (("L1", 1), ("L6", 6), ("L6p", 6), ("L6d", 6)):
        # L6p: next window's encoders + exchange on a side stream; L6d: that, pipelined (results one window late)
        res, tr = run(flower, sharding, window, prefetch=(mode in ("L6p", "L6d")), defer=(mode == "L6d"))
        res["_encoded"] = np.array(tr.sharder.stats["encoded"])
        np.savez(outdir / f"rank{rank}_{mode}.npz", **res)
    if rank == 0:
        np.savez(outdir / "single.npz", **run(flower, False, 1)[0])
    dist.barrier()
    dist.destroy_process_group()
