"""Worker for the multi-rank gloo test of the window-sharded tracker (CPU)."""
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))

import golden_inputs as gi  # noqa: E402
from test_host_logic import StubFlower, T, make_tracker  # noqa: E402

N_FRAMES = 20


class ExchangingFlower:
    """Test double with the feature-exchange interface of RAFTWrapper: a frame's "features" are its id;
    a pair may only be computed from features this rank encoded itself or adopted from a peer.  A window with fewer frames
    than half the ranks is encoded by network: one half (fnet | cnet) per rank."""

    def __init__(self):
        self.features, self.encoded, self.local = {}, 0, 0

    def encode_packed(self, img):
        self.encoded += 1
        return torch.full((6,), float(gi.decode_id(img))), (1, 1)

    def encode_half(self, img, part):
        self.encoded += 0.5
        return torch.full((3,), float(gi.decode_id(img)) if part == 0 else -float(gi.decode_id(img))), (1, 1)

    def adopt_halves(self, frame_id, fbuf, cbuf, img):
        assert int(fbuf[0]) == gi.decode_id(img) == frame_id == -int(cbuf[0])      # the fnet half and the cnet half, in that order
        self.features[frame_id] = torch.cat([fbuf, -cbuf])

    def packed_numel(self, img):
        return 6

    def adopt_packed(self, frame_id, buf, img):
        assert int(buf[0]) == gi.decode_id(img) == frame_id
        self.features[frame_id] = buf

    def reset_cache(self):
        self.features = {}

    def retain(self, frame_ids):
        self.features = {k: v for k, v in self.features.items() if k in set(frame_ids)}

    def compute_pairs(self, pairs):
        out = []
        for lk, limg, rk, rimg in pairs:
            for k, img in ((lk, limg), (rk, rimg)):
                if k not in self.features:              # the start frame, and the frames of a window shorter than the
                    self.local += k != 0                # world size (no exchange: every rank encodes what it needs)
                    self.features[k] = torch.full((6,), float(k))
            flow, occl, sigma = gi.stub_flowou(int(self.features[lk][0]), int(self.features[rk][0]))
            out.append((T(flow), T(occl), T(sigma)))
        return out


class RoundingCache:
    """Flow cache whose entries come back QUANTISED (like the reference's .flowouX16 entries, MFT/utils/io.py:495-563):
    a run that reads an entry chains other values than the run that computed it."""

    def __init__(self):
        self.store, self.hits, self.writes = {}, 0, 0

    def read(self, left_id, right_id):
        if (left_id, right_id) not in self.store:
            return None, None, None
        self.hits += 1
        return tuple(torch.round(t * 64) / 64 for t in self.store[(left_id, right_id)])

    def write(self, left_id, right_id, flow, occl, sigma):
        self.writes += 1
        self.store[(left_id, right_id)] = (flow.clone(), occl.clone(), sigma.clone())


def run(sharded, window, flower, prefetch=False, defer=False, flow_cache=None):
    tr = make_tracker(flower, delta_sharding=sharded)
    tr.init(gi.id_image(0), flow_cache=flow_cache)
    out, i, done = {}, 1, 1            # done: index of the next frame whose meta has not come back yet

    def record(metas):
        nonlocal done
        for m in metas:
            res = m.result
            out[f"flow{done}"] = res.flow.numpy()
            out[f"occl{done}"] = res.occlusion.numpy()
            out[f"sigma{done}"] = res.sigma.numpy()
            done += 1

    while i < N_FRAMES:
        imgs = [gi.id_image(k) for k in range(i, min(i + window, N_FRAMES))]
        nxt = [gi.id_image(k) for k in range(i + window, min(i + 2 * window, N_FRAMES))] if prefetch else None
        if window > 1:
            metas = tr.track_window(imgs, next_imgs=nxt, defer=defer)       # defer: the PREVIOUS window's metas
            assert len(metas) == (len(imgs) if not defer else (0 if i == 1 else window))
        else:
            metas = [tr.track(imgs[0])]
        record(metas)
        i += len(imgs)
        if not defer:
            out[f"chosen{i - 1}"] = tr.last_chosen.numpy()
            out[f"keys{i - 1}"] = np.array(sorted(tr.memory.keys()))
    if defer:
        record(tr.flush_window())
        assert tr.flush_window() == []
    assert done == N_FRAMES
    return out, tr


if __name__ == "__main__":
    outdir = Path(sys.argv[1])
    torch.set_num_threads(2)
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    wx = max(5, dist.get_world_size())       # the feature exchange needs at least one frame per rank in a window
    for mode, window, mk in (("L1", 1, StubFlower), ("L8", 8, StubFlower), ("L1x", 1, ExchangingFlower), ("L5x", wx, ExchangingFlower),
                             ("L5p", wx, ExchangingFlower), ("L5d", wx, ExchangingFlower)):
        fl = mk()
        # L5p: next window's features exchanged early; L5d: that, and every window's results one call late (pipelined)
        res, tr = run(True, window, fl, prefetch=(mode in ("L5p", "L5d")), defer=(mode == "L5d"))
        if mode in ("L1x", "L5x", "L5p", "L5d"):
            st = tr.sharder.stats
            res.update(_encoded=np.array(fl.encoded), _local=np.array(fl.local), _frames=np.array(N_FRAMES - 1), _my_units=np.array(st["my_units"]),
                       _windows=np.array(st["windows"]))
        np.savez(outdir / f"rank{rank}_{mode}.npz", **res)
    # flow cache in the sharded path: a unit is read / written by its owner rank only; a second run over the same
    # frames is served from the owners' caches and chains the QUANTISED entries, exactly like the single-rank tracker
    cache = RoundingCache()
    cold, _ = run(True, 8, StubFlower(), flow_cache=cache)
    fl2 = StubFlower()
    warm, tr2 = run(True, 8, fl2, flow_cache=cache)
    np.savez(outdir / f"rank{rank}_cache.npz", _hits=np.array(cache.hits), _writes=np.array(cache.writes),
             _recomputed=np.array(len([c for c in fl2.calls if c[0] != 0])),
             **{"cold_" + k: v for k, v in cold.items()}, **{"warm_" + k: v for k, v in warm.items()})
    if rank == 0:
        # single-rank reference: per-frame track(); chosen / keys recorded at the same frames as L = 8
        ref, _ = run(False, 1, StubFlower())
        np.savez(outdir / "single.npz", **{k: v for k, v in ref.items() if not k.startswith(("chosen", "keys"))})
        c1 = RoundingCache()
        run(False, 1, StubFlower(), flow_cache=c1)
        ref_warm, _ = run(False, 1, StubFlower(), flow_cache=c1)
        np.savez(outdir / "single_warm.npz", **{k: v for k, v in ref_warm.items() if not k.startswith(("chosen", "keys"))})
    dist.barrier()
    dist.destroy_process_group()
