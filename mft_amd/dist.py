"""Multi-GPU tracking: (frame, delta) units of a look-ahead window sharded over ranks
(the reference has no distributed path at all).

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI; "gloo" in the CPU
tests).  A *unit* is one (left_id -> frame) FlowOU computation.  Units
never depend on tracker state -- RAFT sees only the two images, ``flow_init`` is always None
(``MFT/MFT.py:96-102``) -- so the 7 L units of L consecutive frames are all independent:

  1. window frames are encoded once, by their owner rank (frame j -> rank j mod G), and the
     features (fmap | net | inp, 8.4 MB per 512x512 frame) are all-gathered: no rank re-encodes
     a frame another rank already encoded; a window with fewer frames than half the ranks (the
     per-frame mode, L = 1) is encoded by NETWORK instead: fnet of frame j on rank 2 j, cnet on
     rank 2 j + 1, the two halves meet in the same all-gather;
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


class _NullCtx:
    def __enter__(self): return self
    def __exit__(self, *a): return False


class WindowSharder:
    #: most pairs per engine call; a share is cut into equal batches of at most this many pairs.  fp32 MFMA (64 x 64
    #: tiles, two half-batches on two streams): 2.08 ms per pair at 7 pairs, 2.03 at 8, 1.99 at 14, 1.97 at 16 -> 16.
    #: Split arithmetic: the N = 256 layers run as 128 x 256 tiles, 32 per pair -- 7 or 8 pairs are one round on the 256
    #: CUs, 10 or 11 pairs a full round and a third of one (forced-sharded bench: 88 frames/s with batches of 10 + 11,
    #: against 105 unsharded) -> 8: shares of 21 units run as 7 + 7 + 7.
    MAX_BATCH = 16
    MAX_BATCH_SPLIT = 8

    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("delta_sharding needs an initialised torch.distributed process group")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        self.stats = {"windows": 0, "units": 0, "my_units": 0, "encoded": 0}
        self._prefetch = None          # feature exchange of the next window, started early
        self._pending = None           # a window whose batches and all-gather are under way (track_window(defer=True))

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
    def _start_feature_exchange(self, tracker, frame_ids, imgs, side=False):
        """Encode the window frames this rank owns and start the all-gather of (fmap | net | inp).  With ``side`` the
        encoders and the collective run off the caller's stream (a side stream; the collective is asynchronous), so
        that the NEXT window's features are produced while the current window's RAFT batches run.  Returns a handle
        for ``_finish_feature_exchange`` (None: nothing to exchange)."""
        flower = tracker.flower
        G, L = self.world_size, len(frame_ids)
        if not hasattr(flower, "encode_packed"):
            return None              # a reference-style plugin: everyone encodes what it needs
        halves = 2 * L <= G          # fewer frames than half the ranks (the online mode, L = 1): one NETWORK per rank, see below
        if halves and not hasattr(flower, "encode_half"):
            return None
        stream = None
        if side and hasattr(flower, "ensure_encode_stream"):
            stream = flower.ensure_encode_stream()   # all encoder work of the plugin is serialised on this stream
            if getattr(flower, "_enc_waits_for_device_frames", False) and any(isinstance(im, torch.Tensor) and im.is_cuda for im in imgs):
                stream.wait_stream(torch.cuda.current_stream(tracker.device))   # (device frames may still be being written there)
        ctx = torch.cuda.stream(stream) if stream is not None else _NullCtx()
        if halves:
            # unit u = 2 j + part (part 0: fnet, part 1: cnet) is encoded by rank u: the frame's serial head is ONE encoder,
            # its two halves (N * 256 floats each) meet in the all-gather; ranks >= 2 L send a slot nobody reads
            with ctx:
                numel = flower.packed_numel(imgs[0]) // 2
                if self.rank < 2 * L:
                    send, _ = flower.encode_half(imgs[self.rank // 2], self.rank % 2)
                    send = send.view(1, numel)
                    self.stats["encoded"] += 0.5
                else:
                    send = torch.empty((1, numel), dtype=torch.float32, device=tracker.device)
                recv = torch.empty((G, 1, numel), dtype=send.dtype, device=send.device)
                work = dist.all_gather_into_tensor(recv.view(G, numel), send, group=self.group, async_op=side)
            return dict(ids=list(frame_ids), imgs=list(imgs), recv=recv, send=send, work=work, stream=stream, halves=True)
        slots = -(-L // G)
        mine = [j for j in range(L) if frame_owner(j, G) == self.rank]
        with ctx:
            send = None
            for s_, j in enumerate(mine):
                packed, _ = flower.encode_packed(imgs[j])              # [N * 512]; on the encode stream if there is one
                if send is None:
                    send = torch.empty((slots,) + tuple(packed.shape), dtype=packed.dtype, device=packed.device)
                send[s_] = packed
                self.stats["encoded"] += 1
            if send is None:         # a window shorter than the world: this rank owns none of its frames, its slot is not read
                send = torch.empty((slots, flower.packed_numel(imgs[0])), dtype=torch.float32, device=tracker.device)
            recv = torch.empty((G,) + tuple(send.shape), dtype=send.dtype, device=send.device)
            # (NCCL orders the collective behind the work already on the CURRENT stream -- the side stream here)
            work = dist.all_gather_into_tensor(recv.view(G * send.shape[0], *send.shape[1:]), send, group=self.group,
                                               async_op=side)
        return dict(ids=list(frame_ids), imgs=list(imgs), recv=recv, send=send, work=work, stream=stream, halves=False)

    def _finish_feature_exchange(self, tracker, h):
        """Wait for the exchange and install the features in the flow plugin's per-frame cache (all ranks)."""
        if h is None:
            return
        on_gpu = h["recv"].is_cuda
        cur = torch.cuda.current_stream(h["recv"].device) if on_gpu else None
        if h["stream"] is not None:
            # the SIDE stream waits for the collective and marks the moment the features are complete with an event of its own:
            # the caller's stream waits for that event -- and so can any other stream that consumes the features (the plugin's
            # lanes, flow_config.frames_in_flight) without waiting for whatever else is queued on the caller's stream
            with torch.cuda.stream(h["stream"]):
                if h["work"] is not None:
                    h["work"].wait()
                ready = h["stream"].record_event()
            cur.wait_event(ready)
            h["recv"].record_stream(cur)
        else:
            if h["work"] is not None:
                h["work"].wait()                      # the caller's stream now waits for the collective
            ready = cur.record_event() if on_gpu else None
        G = self.world_size
        kw = {"ready": ready} if ready is not None else {}       # (plugins without streams: the plain interface)
        for j, fid in enumerate(h["ids"]):
            if h["halves"]:
                tracker.flower.adopt_halves(fid, h["recv"][2 * j, 0], h["recv"][2 * j + 1, 0], h["imgs"][j], **kw)
            else:
                tracker.flower.adopt_packed(fid, h["recv"][frame_owner(j, G), j // G], h["imgs"][j], **kw)

    # ------------------------------------------------------------------ the window
    def track_window(self, tracker, imgs, next_imgs=None, defer=False):
        """Track ``imgs`` (the next L frames); returns their metas in order.  Identical results and
        tracker state on every rank.  ``next_imgs``: the frames of the FOLLOWING window, if known -- their
        encoding and feature exchange are started on a side stream now and overlap this window's RAFT batches.

        ``defer``: pipelined mode -- this window's RAFT batches are enqueued and its FlowOU all-gather is started
        (asynchronously), but its chain + select is left pending; what is returned are the metas of the window
        deferred by the PREVIOUS call ([] the first time), ``flush`` hands back the last one.  The all-gather and the
        replicated selections of window w then overlap the RAFT batches of window w + 1 instead of standing between
        two windows' batches.  Same results, one window later."""
        if not defer and self._pending is not None:
            # (checked BEFORE anything is enqueued: a rank that raised after issuing collectives would leave its peers hanging)
            raise RuntimeError("flush() the deferred window before switching to undeferred calls")
        cur = self._start_window(tracker, imgs, next_imgs, async_gather=defer)
        if not defer:
            return self._finish_window(tracker, cur)
        prev, self._pending = self._pending, cur
        return self._finish_window(tracker, prev) if prev is not None else []

    def flush(self, tracker):
        """Metas of the window a ``defer`` call left pending ([] if there is none)."""
        prev, self._pending = self._pending, None
        return self._finish_window(tracker, prev) if prev is not None else []

    def _start_window(self, tracker, imgs, next_imgs, async_gather):
        G, r = self.world_size, self.rank
        L = len(imgs)
        d = tracker.time_direction
        pend = self._pending
        last_id = pend["frame_ids"][-1] if pend is not None else tracker.current_frame_i
        frame_ids = [last_id + d * (j + 1) for j in range(L)]
        window_img = dict(zip(frame_ids, imgs))
        if pend is not None:                       # frames of the deferred window are not in tracker.memory yet
            window_img.update(zip(pend["frame_ids"], pend["imgs"]))

        def img_of(fid):
            return window_img[fid] if fid in window_img else tracker.memory[fid]['img']

        plans = [tracker._plan(fid) for fid in frame_ids]
        units = [(j, k) for j, plan in enumerate(plans) for k in range(len(plan))]
        shares = split_units(len(units), G)
        off, cnt = shares[r]
        slots = max(c for _, c in shares)

        keep = set(frame_ids) | (set(pend["frame_ids"]) if pend is not None else set())
        tracker._window_ids = set(keep)
        pre = self._prefetch
        self._prefetch = None
        if pre is not None and pre["ids"] == frame_ids:
            self._finish_feature_exchange(tracker, pre)                      # started during the previous window
        else:
            # (with several frames in flight the exchange of THIS window, too, runs on the plugin's encode stream: on the caller's
            # stream its encoders would queue behind the previous window's selection, i.e. behind the previous window's whole batch --
            # the per-frame mode, which has nothing to prefetch, then never has two frames in flight)
            lanes = int(getattr(tracker.flower, "_fif", 1)) > 1 and hasattr(tracker.flower, "ensure_encode_stream")
            self._finish_feature_exchange(tracker, self._start_feature_exchange(tracker, frame_ids, imgs, side=lanes))
        if next_imgs:
            nxt_ids = [frame_ids[-1] + d * (j + 1) for j in range(len(next_imgs))]
            tracker._window_ids |= set(nxt_ids)
            self._prefetch = self._start_feature_exchange(tracker, nxt_ids, list(next_imgs), side=True)

        # ---- my share, in engine batches
        H, W = tracker.img_H, tracker.img_W
        dev = tracker.device
        # FlowOU travels in the packed per-pixel format [H, W, 4] = (fx, fy, occl, sigma): the native engine
        # writes it straight into the send buffer and the chain kernel gathers 16 bytes per tap from the
        # receive buffer -- no staging copies on either side
        send = torch.empty(max(slots, 1), H, W, 4, dtype=torch.float32, device=dev)
        mine = units[off: off + cnt]
        max_batch = self.MAX_BATCH_SPLIT if getattr(tracker.flower, "_arith", 0) == 1 else self.MAX_BATCH
        # Flow cache (MFT/MFT.py:96-102, 214-219): a unit is looked up -- and, when computed, written back -- by its OWNER
        # rank only, so hit or miss cannot differ between ranks: whatever the owner puts into its send slot (the cached,
        # quantised FlowOU or the fresh one) is what every rank chains.  Any read error means recompute.
        cache = getattr(tracker, "flow_cache", None)
        todo = list(range(cnt))                     # positions in `mine` that go through the engine
        if cache is not None:
            todo = []
            for i, (j, k) in enumerate(mine):
                _, left_id, use_cache = plans[j][k]
                hit = None
                if use_cache:
                    try:
                        f, o, s_ = cache.read(left_id, frame_ids[j])
                        assert f is not None
                        hit = (torch.as_tensor(f).to(dev), torch.as_tensor(o).to(dev), torch.as_tensor(s_).to(dev))
                    except Exception:
                        hit = None
                if hit is None:
                    todo.append(i)
                else:
                    send[i].copy_(torch.cat([hit[0].reshape(2, H, W), hit[1].reshape(1, H, W), hit[2].reshape(1, H, W)], 0).permute(1, 2, 0))
                    self.stats["cache_hits"] = self.stats.get("cache_hits", 0) + 1
        n_todo = len(todo)
        n_batches = -(-n_todo // max_batch) if n_todo else 0
        bounds = [(n_todo * i) // n_batches for i in range(n_batches + 1)] if n_todo else [0]
        # (no cache, or nothing found: the engine writes into the send buffer itself -- unless the plugin keeps several frames in
        # flight (flow_config.frames_in_flight): then the batch rides a lane, on that lane's stream into that lane's memory, so that
        # consecutive windows' batches overlap like consecutive frames of the single-GPU tracker, and its results are copied into
        # the send buffer behind the lane's event: 4 MB per unit, microseconds against the batch's milliseconds)
        contiguous = todo == list(range(cnt)) and int(getattr(tracker.flower, "_fif", 1)) <= 1
        for b0, b1 in zip(bounds[:-1], bounds[1:]):
            batch = [mine[i] for i in todo[b0: b1]]
            pairs = []
            for j, k in batch:
                left_id = plans[j][k][1]
                pairs.append((left_id, img_of(left_id), frame_ids[j], imgs[j]))
            want_planes = cache is not None and any(plans[j][k][2] for j, k in batch)
            res = tracker._flows_for_pairs(pairs, packed_out=send[b0: b0 + len(batch)] if contiguous else True,
                                           planar=want_planes)
            for s_, out in enumerate(res):
                i = todo[b0 + s_]
                if len(out) < 4:                      # a plugin without packed output: interleave here
                    send[i].copy_(torch.cat([out[0], out[1], out[2]], 0).permute(1, 2, 0))
                elif not contiguous:
                    send[i].copy_(out[3])
                j, k = mine[i]
                if cache is not None and plans[j][k][2]:
                    cache.write(plans[j][k][1], frame_ids[j], out[0], out[1], out[2])
        recv = torch.empty((G,) + tuple(send.shape), dtype=send.dtype, device=send.device)     # [G, slots, H, W, 4]
        work = dist.all_gather_into_tensor(recv.view(G * send.shape[0], *send.shape[1:]), send, group=self.group,
                                           async_op=async_gather)
        self.stats["windows"] += 1
        self.stats["units"] += len(units)
        self.stats["my_units"] += cnt
        return dict(frame_ids=frame_ids, imgs=list(imgs), plans=plans, shares=shares, send=send, recv=recv,
                    work=work if async_gather else None)

    def _finish_window(self, tracker, w):
        """Replicated chain + select of a started window, frame by frame -> its metas."""
        if w["work"] is not None:
            w["work"].wait()                          # the caller's stream waits for the collective (no host wait)
        frame_ids, imgs, plans, recv = w["frame_ids"], w["imgs"], w["plans"], w["recv"]
        owner_slot = {}
        for rr, (o, c) in enumerate(w["shares"]):
            for s_ in range(c):
                owner_slot[o + s_] = (rr, s_)
        later = set(self._pending["frame_ids"]) if self._pending is not None else set()
        if self._prefetch is not None:
            later |= set(self._prefetch["ids"])
        metas, u = [], 0
        L = len(frame_ids)
        for j, fid in enumerate(frame_ids):
            plan = plans[j]
            lefts, rights = [], []
            for k in range(len(plan)):
                rr, s_ = owner_slot[u]
                u += 1
                rights.append(recv[rr, s_])            # packed [H, W, 4]
                lefts.append(tracker.memory[plan[k][1]]['result'].planes())
            # features stay for: the rest of this window, a window already started behind it, a prefetched one
            tracker._window_ids = set(frame_ids[j + 1:]) | later
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
