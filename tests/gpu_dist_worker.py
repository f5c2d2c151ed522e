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


def run(flower, sharding, window, prefetch=False, defer=False, size=(128, 160), n_frames=N_FRAMES, flow_cache=None, seed=4):
    c = Config()
    c.deltas = [np.inf, 1, 2, 4, 8]
    c.occlusion_threshold = 0.02
    c.delta_sharding = sharding
    c.flow_config = Config()
    c.flow_config.of_class = lambda cfg: flower
    tr = MFT(c)
    vid = SyntheticVideo(size[0], size[1], n_frames=n_frames, seed=seed)
    tr.init(vid[0], flow_cache=flow_cache)
    out, i, done = {}, 1, 1

    def record(metas):
        nonlocal done
        for m in metas:
            res = m.result
            out[f"flow{done}"], out[f"occl{done}"], out[f"sigma{done}"] = \
                res.flow.numpy(), res.occlusion.numpy(), res.sigma.numpy()
            done += 1

    while i < n_frames:
        imgs = [vid[k] for k in range(i, min(i + window, n_frames))]
        nxt = [vid[k] for k in range(i + window, min(i + 2 * window, n_frames))] if prefetch else None
        record(tr.track_window(imgs, next_imgs=nxt, defer=defer) if window > 1 else [tr.track(imgs[0])])
        i += len(imgs)
    if defer:
        record(tr.flush_window())
    assert done == n_frames
    out["final_chosen"] = tr.last_chosen.cpu().numpy()
    out["final_keys"] = np.array(sorted(tr.memory.keys()))
    return out, tr


if __name__ == "__main__":
    outdir = Path(sys.argv[1])
    # MFT_DIST_BACKEND=gloo + MFT_DIST_ONE_GPU=1: every rank on cuda:0, collectives on DEVICE tensors through gloo (RCCL refuses
    # two ranks on one device) -- the real plugin, engines, side streams and asynchronous collectives with world_size > 1 on the
    # one GPU a test box has
    backend = os.environ.get("MFT_DIST_BACKEND", "nccl")
    one_gpu = os.environ.get("MFT_DIST_ONE_GPU") == "1"
    local_rank = 0 if one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    fc = Config()
    fc.flow_iters = 4
    flower = RAFTWrapper(fc, state_dict=make_weights(7))
    rank, world = dist.get_rank(), dist.get_world_size()
    sharding = "force" if world == 1 else True
    modes = [("L1", 1), ("L6", 6), ("L6p", 6), ("L6d", 6)]
    if world > 1:
        modes.append(("LG", world))                         # one frame per rank and window: the bench's default window
    for mode, window in modes:
        # L6p: next window's encoders + exchange on a side stream; L6d: that, pipelined (results one window late)
        res, tr = run(flower, sharding, window, prefetch=(mode in ("L6p", "L6d")), defer=(mode == "L6d"))
        res["_encoded"] = np.array(tr.sharder.stats["encoded"])
        np.savez(outdir / f"rank{rank}_{mode}.npz", **res)
    if rank == 0:
        np.savez(outdir / "single.npz", **run(flower, False, 1)[0])
    if world > 1:
        # the flow cache in the sharded path (HBM tier: exact entries): cold and warm runs equal the uncached tracker
        from mft_amd.io import FlowCache
        cache = FlowCache(None, max_GPU_RAM_MB=4000)
        cold, _ = run(flower, sharding, world, prefetch=True, defer=True, flow_cache=cache)
        n_cold = cache.n_saved
        warm, tr = run(flower, sharding, world, prefetch=True, defer=True, flow_cache=cache)
        np.savez(outdir / f"rank{rank}_cachecold.npz", **cold)
        np.savez(outdir / f"rank{rank}_cachewarm.npz", _hits=np.array(tr.sharder.stats.get("cache_hits", 0)), _cold_writes=np.array(n_cold),
                 _warm_writes=np.array(cache.n_saved - n_cold), **warm)
        # one 512 x 512 window at the production settings (12 iterations, split arithmetic, pinned tile kernels) + a ragged
        # one-frame window (the by-network encode) behind it
        fc2 = Config()
        fc2.flow_iters = 12
        big = RAFTWrapper(fc2, state_dict=make_weights(7))
        res, _ = run(big, sharding, world, prefetch=True, defer=True, size=(512, 512), n_frames=world + 2, seed=9)
        np.savez(outdir / f"rank{rank}_big.npz", **res)
        res1, _ = run(big, sharding, 1, size=(512, 512), n_frames=4, seed=9)
        np.savez(outdir / f"rank{rank}_big1.npz", **res1)
        if rank == 0:
            np.savez(outdir / "single_big.npz", **run(big, False, 1, size=(512, 512), n_frames=world + 2, seed=9)[0])
            np.savez(outdir / "single_big1.npz", **run(big, False, 1, size=(512, 512), n_frames=4, seed=9)[0])
    if world > 1 and os.environ.get("MFT_DIST_1080P") == "1":
        # BASELINE config 5's multi-GPU half on the one GPU of a test box: 1080 x 1920 windows (136 x 240 cells: the HBM-resident
        # pyramid, 272 workgroups and more per kernel) sharded over the ranks, pipelined, against the single-process tracker
        fc3 = Config()
        fc3.flow_iters = 12
        hd = RAFTWrapper(fc3, state_dict=make_weights(7))
        res, _ = run(hd, sharding, world, prefetch=True, defer=True, size=(1080, 1920), n_frames=2 * world + 1, seed=11)
        np.savez(outdir / f"rank{rank}_hd.npz", **res)
        if rank == 0:
            np.savez(outdir / "single_hd.npz", **run(hd, False, 1, size=(1080, 1920), n_frames=2 * world + 1, seed=11)[0])
    dist.barrier()
    dist.destroy_process_group()
