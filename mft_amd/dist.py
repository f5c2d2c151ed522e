"""Multi-GPU tracking: (frame, delta) units of a look-ahead window sharded over ranks
(the reference has no distributed path at all).

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI; "gloo" in the CPU
tests).  A *unit* is one (left_id -> frame) FlowOU computation.  Units
never depend on tracker state -- RAFT sees only the two images, ``flow_init`` is always None
(``MFT/MFT.py:96-102``) -- so the 7 L units of L consecutive frames are all independent:

  1. window frames are encoded once, by their owner rank (frame j -> rank j mod G), and the
     features (fmap | net | inp, 8.4 MB per 512x512 frame) are all-gathered: no rank re-encodes
     a frame another rank already encoded;
  2. the units, in (frame, selection-order) order, are cut into G contiguous, equally sized
     shares (sizes differ by at most one); every rank runs its share through the native RAFT
     engine in equal batches of up to 16 pairs -- at or above the single-GPU tracker's 7, where the
     conv GEMMs run at their full rate -- whatever G is;
  3. ONE all-gather of the raw FlowOU results in the packed per-pixel format (fx, fy, occl, sigma:
     16 B per pixel and unit; written into the send buffer by the engine itself; the last slot of a
     short share is simply not read -- nothing is zero-filled, nothing is staged);
  4. every rank runs the fused chain + select kernel for the window's frames IN FRAME ORDER
     (frame t's chains read ``memory[t - delta]``), so ``tracker.memory`` stays replicated and
     bitwise equal on all ranks, and equal to the single-GPU tracker: the unit results do not
     depend on how they are batched (``test_batch_invariance_bitwise``) and chain + select is the
     very kernel the single-GPU path runs.

L = 1 is the online mode (``MFT.track``): the <= 7 units of the current frame are split over the
ranks, which caps the speed-up at the per-rank batch efficiency (a share of one pair runs the GEMMs
at about half their rate).  L >= G is the offline mode (``MFT.track_window``): every rank always
has full batches, results come back L frames at a time.

Payloads per window at 512x512: features L x 8.4 MB, FlowOU 7 L x 4.19 MB (L = 16: 134 MB +
470 MB gathered, i.e. 17 + 59 MB sent per rank at G = 8) against ~16 ms of RAFT per 7 units --
two collectives per window, not a bucketed/overlapped pipeline.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
import torch.distributed as dist


def split_units(n_units: int, world_size: int):
    """Contiguous, balanced shares: [(offset, count)] per rank; counts differ by at most one."""
    base, extra = divmod(n_units, world_size)
    out, off = [], 0
    for r in range(world_size):
        cnt = base + (1 if r < extra else 0)
        out.append((off, cnt))
        off += cnt
    return out


def frame_owner(j: int, world_size: int) -> int:
    """Rank that encodes the j-th frame of a window."""
    return j % world_size


class WindowSharder:
    #: most pairs per engine call.  Per-pair time of a call (512 x 512, 12 iterations, two half-batches on two
    #: streams): 2.08 ms at 7 pairs, 2.03 at 8, 1.99 at 14, 1.97 at 16 (the 64-row tiles of 8 k pairs divide the 256 CUs
    #: evenly for every layer) -- a share is cut into equal batches of at most this many pairs
    MAX_BATCH = 16

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("delta_sharding needs an initialised torch.distributed process group")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        self.stats = {"windows": 0, "units": 0, "my_units": 0, "encoded": 0}

    @classmethod
    def from_environment(cls):
        return cls()

    # ------------------------------------------------------------------ collectives
    def _all_gather(self, send: torch.Tensor) -> torch.Tensor:
        """[S, ...] per rank -> [G, S, ...] (rank-major)."""
        recv = torch.empty((self.world_size,) + tuple(send.shape), dtype=send.dtype, device=send.device)
        dist.all_gather_into_tensor(recv.view(self.world_size * send.shape[0], *send.shape[1:]), send,
                                    group=self.group)
        return recv

    # ------------------------------------------------------------------ features
    def _exchange_features(self, tracker, frame_ids, imgs):
        """Encode each window frame on its owner and all-gather (fmap | net | inp); afterwards the
        flow plugin's per-frame cache holds every window frame on every rank."""
        flower = tracker.flower
        G, L = self.world_size, len(frame_ids)
        if not hasattr(flower, "encode_packed"):
            return                                   # reference-style plugin: nothing to exchange
        if L < G:
            return                                   # fewer frames than ranks: everyone encodes what it needs
        slots = -(-L // G)
        mine = [j for j in range(L) if frame_owner(j, G) == self.rank]
        send = None
        for s, j in enumerate(mine):
            packed, geom = flower.encode_packed(imgs[j])          # [N, 512]
            if send is None:
                send = torch.empty((slots,) + tuple(packed.shape), dtype=packed.dtype, device=packed.device)
            send[s] = packed
            self.stats["encoded"] += 1
        if send is None:                              # cannot happen for L >= G, kept for clarity
            packed, geom = flower.encode_packed(imgs[0])
            send = torch.empty((slots,) + tuple(packed.shape), dtype=packed.dtype, device=packed.device)
        recv = self._all_gather(send)                 # [G, slots, N, 512]
        for j in range(L):
            flower.adopt_packed(frame_ids[j], recv[frame_owner(j, G), j // G], imgs[j])

    # ------------------------------------------------------------------ the window
    def track_window(self, tracker, imgs):
        """Track ``imgs`` (the next L frames); returns their metas in order.  Identical results and
        tracker state on every rank."""
        G, r = self.world_size, self.rank
        L = len(imgs)
        d = tracker.time_direction
        frame_ids = [tracker.current_frame_i + d * (j + 1) for j in range(L)]
        window_img = dict(zip(frame_ids, imgs))

        def img_of(fid):
            return window_img[fid] if fid in window_img else tracker.memory[fid]['img']

        plans = [tracker._plan(fid) for fid in frame_ids]
        units = [(j, k) for j, plan in enumerate(plans) for k in range(len(plan))]
        shares = split_units(len(units), G)
        off, cnt = shares[r]
        slots = max(c for _, c in shares)

        tracker._window_ids = set(frame_ids)
        self._exchange_features(tracker, frame_ids, imgs)

        # ---- my share, in engine batches
        H, W = tracker.img_H, tracker.img_W
        dev = tracker.device
        # FlowOU travels in the packed per-pixel format [H, W, 4] = (fx, fy, occl, sigma): the native engine
        # writes it straight into the send buffer and the chain kernel gathers 16 bytes per tap from the
        # receive buffer -- no staging copies on either side
        send = torch.empty(max(slots, 1), H, W, 4, dtype=torch.float32, device=dev)
        mine = units[off: off + cnt]
        n_batches = -(-cnt // self.MAX_BATCH) if cnt else 0
        bounds = [(cnt * i) // n_batches for i in range(n_batches + 1)] if cnt else [0]
        for b0, b1 in zip(bounds[:-1], bounds[1:]):
            batch = mine[b0: b1]
            pairs = []
            for j, k in batch:
                left_id = plans[j][k][1]
                pairs.append((left_id, img_of(left_id), frame_ids[j], imgs[j]))
            res = tracker._flows_for_pairs(pairs, packed_out=send[b0: b0 + len(batch)], planar=False)
            for s, r in enumerate(res):
                if len(r) < 4:                        # a plugin without packed output: interleave here
                    send[b0 + s].copy_(torch.cat([r[0], r[1], r[2]], 0).permute(1, 2, 0))
        recv = self._all_gather(send)                 # [G, slots, H, W, 4]
        self.stats["windows"] += 1
        self.stats["units"] += len(units)
        self.stats["my_units"] += cnt

        owner_slot = {}
        for rr, (o, c) in enumerate(shares):
            for s in range(c):
                owner_slot[o + s] = (rr, s)

        # ---- replicated chain + select, frame by frame
        metas, u = [], 0
        for j, fid in enumerate(frame_ids):
            plan = plans[j]
            lefts, rights = [], []
            for k in range(len(plan)):
                rr, s = owner_slot[u]
                u += 1
                rights.append(recv[rr, s])            # packed [H, W, 4]
                lefts.append(tracker.memory[plan[k][1]]['result'].planes())
            if j == L - 1:
                tracker._window_ids = set()
            metas.append(tracker._finish_frame(fid, imgs[j], plan, lefts, rights))
        return metas


# the per-frame mode of round 1 is the L = 1 window
DeltaSharder = WindowSharder


def shard_indices(K: int, world_size: int, rank: int):
    """Units of ``rank`` among the K units of one frame (L = 1): a contiguous share."""
    off, cnt = split_units(K, world_size)[rank]
    return list(range(off, off + cnt))


def make_meta(result):
    return SimpleNamespace(result=result)
