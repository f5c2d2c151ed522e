"""MFT tracker -- drop-in for ``MFT/MFT.py:13-239``.

Same public surface: ``MFT(config)``, ``.init(img, start_frame_i=0,
time_direction=1, flow_cache=None) -> meta``, ``.track(img) -> meta`` with
``meta.result`` a ``FlowOUTrackingResult`` on the CPU, ``tracker.C`` (monkey-
patched by runners), ``tracker.memory[frame_i] = {'img', 'result'}`` on the
device, plus the module-level ``get_flowou_with_cache`` and ``chain_results``.

What changed underneath (same results):
  * the <= 7 (left -> current) flow pairs of a frame are gathered first and run
    as ONE batched pass through the native RAFT engine (``compute_flow_many``)
    instead of 7 separate forward passes; frames are encoded once and cached;
  * the 7 x ``chain_results`` + stack / max / gather selection is one fused HIP
    kernel (``mftx_chain_select``);
  * with ``torch.distributed`` initialised and ``C.delta_sharding`` set, the
    (frame, delta) flow computations of a frame (``track``) or of a look-ahead
    window of frames (``track_window``) are sharded over ranks and reassembled
    with an all-gather (``mft_amd/dist.py``).
Python remains the owner of the delta bookkeeping, the memory ring and the cache.
"""
from __future__ import annotations

import logging
from types import SimpleNamespace

import numpy as np
import torch

from . import ops
from .results import FlowOUTrackingResult, PendingHostResult

logger = logging.getLogger(__name__)


def is_packed(r):
    """A right operand in the packed per-pixel format [H, W, 4] (fx, fy, occl, sigma)."""
    return isinstance(r, torch.Tensor)


def pack_planes(planes):
    """(flow[2,H,W], occl[1,H,W], sigma[1,H,W]) -> packed [H, W, 4] (one strided copy)."""
    f, o, s_ = planes
    return torch.cat([f.reshape(2, *f.shape[-2:]), o.reshape(1, *o.shape[-2:]), s_.reshape(1, *s_.shape[-2:])], 0).permute(1, 2, 0).contiguous()


def unpack_planes(r):
    """packed [H, W, 4] -> (flow[2,H,W], occl[1,H,W], sigma[1,H,W]) (contiguous copies)."""
    p = r.permute(2, 0, 1)
    return p[0:2].contiguous(), p[2:3].contiguous(), p[3:4].contiguous()


class HipBackend:
    """chain / select on libmftx (the only product backend)."""

    @staticmethod
    def chain(L, R):
        return ops.chain(L, R)

    @staticmethod
    def select(cands, thr):
        return ops.select(cands, thr, want_chosen=True)

    @staticmethod
    def chain_select(Ls, Rs, thr):
        """Rs: per candidate either the (flow, occl, sigma) planes or the packed [H, W, 4] tensor."""
        if all(is_packed(r) for r in Rs):
            return ops.chain_select_packed(Ls, Rs, thr, want_chosen=True)
        return ops.chain_select(Ls, [unpack_planes(r) if is_packed(r) else r for r in Rs], thr, want_chosen=True)


class MFT():
    def __init__(self, config, backend=None, device='cuda'):
        """config: a mft_amd.config.Config, e.g. from configs/MFT_cfg.py."""
        self.C = config   # must be named self.C, runners monkey-patch it
        self.flower = config.flow_config.of_class(config.flow_config)
        self.device = device
        self.backend = backend if backend is not None else HipBackend()
        self.sharder = None

    # ------------------------------------------------------------------ init
    def init(self, img, start_frame_i=0, time_direction=1, flow_cache=None, **kwargs):
        """Initialise on the first frame (MFT/MFT.py:22-53)."""
        self.img_H, self.img_W = img.shape[:2]
        self.start_frame_i = start_frame_i
        self.current_frame_i = self.start_frame_i
        assert time_direction in [+1, -1]
        self.time_direction = time_direction
        self.flow_cache = flow_cache
        if hasattr(self.flower, "reset_cache"):
            self.flower.reset_cache()
        if hasattr(self.flower, "set_nominal_pairs"):      # kernel choices are made for the steady-state batch, not per call
            self.flower.set_nominal_pairs(len(set(self.C.deltas)) if self.C.deltas else 1)
        # a private copy: the caller may recycle its buffer (a pinned staging ring, mft_amd/video.py: FrameRing), and the
        # template stays in memory for the whole sequence when inf is among the deltas
        self.template_img = img.copy() if hasattr(img, "copy") else img.clone()
        self.memory = {
            self.start_frame_i: {
                'img': self.template_img,
                'result': FlowOUTrackingResult.identity((self.img_H, self.img_W), device=self.device)
            }
        }
        self.last_pairs = []
        self.last_chosen = None
        self._window_ids = set()
        if self.C.delta_sharding:
            from .dist import WindowSharder
            self.sharder = WindowSharder.from_environment()
        meta = SimpleNamespace()
        meta.result = self.memory[self.start_frame_i]['result'].clone().cpu()
        return meta

    # ----------------------------------------------------------------- track
    def _plan(self, frame_i=None):
        """Delta bookkeeping of MFT/MFT.py:74-91 for frame ``frame_i`` (default: the current one):
        [(delta, left_id, use_cache)] in selection order."""
        if frame_i is None:
            frame_i = self.current_frame_i
        plan, used = [], []
        for delta in self.C.deltas:
            left_id = frame_i - delta * self.time_direction
            if self.is_before_start(left_id):
                if np.isinf(delta):
                    left_id = self.start_frame_i
                else:
                    continue
            left_id = int(left_id)
            if left_id in used:
                continue
            used.append(left_id)
            use_cache = bool(np.isfinite(delta) or self.C.cache_delta_infinity)
            plan.append((delta, left_id, use_cache))
        # selection order: inf first, then ascending delta (MFT/MFT.py:114)
        plan.sort(key=lambda e: 0 if np.isinf(e[0]) else e[0])
        return plan

    def _sharded(self):
        return self.sharder is not None and (self.sharder.world_size > 1 or self.C.delta_sharding == "force")

    def track(self, input_img, debug=False, **kwargs):
        """Track one frame (MFT/MFT.py:55-154)."""
        if self._sharded():
            return self.sharder.track_window(self, [input_img])[0]
        frame_i = self.current_frame_i + self.time_direction
        plan = self._plan(frame_i)
        rights = self._flows_for(plan, frame_i, input_img)
        lefts = [self.memory[left_id]['result'].planes() for _, left_id, _ in plan]
        return self._finish_frame(frame_i, input_img, plan, lefts, rights)

    def track_window(self, imgs, next_imgs=None, defer=False):
        """Track the next ``len(imgs)`` frames and return their metas in order.  Same results as calling
        ``track`` on each; with ``C.delta_sharding`` and several ranks the (frame, delta) flow
        computations of the whole window are sharded over the GPUs (``mft_amd/dist.py``).  ``next_imgs``
        (optional): the frames of the window after this one, so that their encoding can start early.
        ``defer`` (multi-GPU only; ignored otherwise): pipelined mode -- returns the metas of the window deferred by
        the previous call ([] the first time) while this window's result exchange overlaps the next window's flow
        batches; ``flush_window()`` returns the last one."""
        imgs = list(imgs)
        if self._sharded() and imgs:
            return self.sharder.track_window(self, imgs, next_imgs=list(next_imgs) if next_imgs else None, defer=defer)
        return [self.track(img) for img in imgs]

    def flush_window(self):
        """Metas of the window a ``track_window(..., defer=True)`` call left pending ([] if none)."""
        return self.sharder.flush(self) if self._sharded() else []

    def _finish_frame(self, frame_i, input_img, plan, lefts, rights):
        """Chain every candidate onto its stored (template -> left) result, pick the best per pixel
        (MFT/MFT.py:104-143), store the frame and clean the ring."""
        meta = SimpleNamespace()
        # C.timers_enabled (MFT/MFT.py:73, 104-113, 143): the reference times the chains and the selection with CUDA events
        # and logs them at DEBUG level; here both are ONE kernel, timed and logged as such
        timed = bool(self.C.timers_enabled) and torch.cuda.is_available()
        if timed:
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
        flow, occl, sigma, chosen = self.backend.chain_select(lefts, rights, self.C.occlusion_threshold)
        if timed:
            t1.record()
            torch.cuda.synchronize()
            logger.debug("chain + selection (%d candidates, one kernel): %.2fms", len(lefts), t0.elapsed_time(t1))
        # invalid flows are already marked occluded inside the selection kernel
        result = FlowOUTrackingResult(flow, occl, sigma, validate=False)
        lazy = self._lazy_host_results() and flow.is_cuda and not self.C.keep_result_on_device
        if self.C.keep_result_on_device:
            meta.result = result.clone()       # a copy: the consumer may move it in place (meta.result.cpu())
        elif lazy:
            meta.result = self._host_result_async(result)
        else:
            meta.result = result.clone().cpu()
        # the guard runs BEFORE the tracker's frame counter, memory ring and last_pairs change.  (The frame's flows are in the
        # flow cache by then and, on the unsynchronised paths, up to nonfinite_check_every earlier frames were stored unchecked:
        # after a FloatingPointError re-initialise the tracker -- with raft_params.arith = 'fp32' in the configuration if the
        # message names the split arithmetic -- rather than calling track() again.)
        self._check_nonfinite(synced=not (self.C.keep_result_on_device or lazy))
        self.current_frame_i = frame_i
        self.last_pairs = [(left_id, frame_i) for _, left_id, _ in plan]
        self.last_chosen = chosen
        self.memory[frame_i] = {'img': input_img, 'result': result}
        self.cleanup_memory()
        return meta

    def _lazy_host_results(self):
        """C.lazy_host_result (default True): ``meta.result`` of ``track()`` is a ``PendingHostResult`` -- a CPU result whose planes
        arrive by an asynchronous copy and whose first access waits for them -- instead of a synchronous ``.cpu()`` in every call.
        False restores the blocking copy (pageable host tensors, as the reference allocates them)."""
        v = self.C.lazy_host_result
        return v if isinstance(v, bool) else True

    def _host_result_async(self, result):
        """The frame's result -> one pinned host buffer [4, H, W] by the copy kernel, on the caller's stream right behind the selection
        kernel (no SDMA queue: mft_amd/video.py ResultDrain found a pinned download there holding back the pinned uploads
        queued behind it).  The buffer comes from torch's caching pinned allocator: results the caller drops are recycled, results it
        keeps stay pinned (4.2 MB each at 512 x 512).  The tracker holds on to every buffer until its copy has run, so that a
        result dropped unread cannot be recycled -- by this or any other user of the pinned pool -- under the copy that fills it."""
        host = torch.empty((4, self.img_H, self.img_W), dtype=torch.float32, pin_memory=True)
        for dst, src in ((host[0:2], result.flow), (host[2:3], result.occlusion), (host[3:4], result.sigma)):
            ops.copy_bytes(src.contiguous(), dst)
        check = words = None
        every = self.C.nonfinite_check_every
        guard_off = ((isinstance(every, (int, float)) and not isinstance(every, bool) and every <= 0) or
                     (isinstance(self.C.raise_on_nonfinite, bool) and not self.C.raise_on_nonfinite))
        if hasattr(self.flower, "nonfinite_snapshot") and not guard_off:
            n = len(self.flower._all_engines())
            words = torch.zeros((n, 4), dtype=torch.int32, pin_memory=True)
            self.flower.nonfinite_snapshot(words)
            flower = self.flower

            def check(words=words, flower=flower):
                bad = int(words[:, 0].sum())
                if bad:
                    raise flower.nonfinite_error(bad)
        ev = torch.cuda.current_stream().record_event()
        pending = getattr(self, "_pending_host", None)
        if pending is None:
            pending = self._pending_host = []
        while pending and pending[0][0].query():
            pending.pop(0)
        pending.append((ev, host, words))
        while len(pending) > 64:                  # (a caller that never reads: bound the list, the oldest copy ran long ago)
            pending.pop(0)[0].synchronize()
        return PendingHostResult(host, ev, on_wait=check)

    def _check_nonfinite(self, synced):
        """The flow plugin counts, on the device, output pixels the reference could not have produced from finite activations: a
        NaN in any plane or an infinite flow / occlusion (mftx_raft_set_nonfinite_counter; sigma = +inf, which MFT/raft.py:62
        produces for log-variances > ~88.7 and the reference tolerates, is NOT counted).  The count is read -- and a
        FloatingPointError raised, the tracker still at frame t-1 -- where the host has synchronised anyway (the result just went
        to the CPU), otherwise every C.nonfinite_check_every frames (default 32).  The reference has no such guard
        (MFT/MFT.py:96-143), so it can be switched off entirely: C.nonfinite_check_every = 0 or C.raise_on_nonfinite = False
        disable it on BOTH paths (check_nonfinite() / ResultDrain(nonfinite_from=...) remain for explicit use)."""
        if not hasattr(self.flower, "raise_if_nonfinite"):
            return
        every = self.C.nonfinite_check_every
        every = int(every) if isinstance(every, (int, float)) and not isinstance(every, bool) else 32   # (unset attribute: a falsy Config)
        off = every <= 0 or (isinstance(self.C.raise_on_nonfinite, bool) and not self.C.raise_on_nonfinite)
        if off:
            return
        if synced or not hasattr(self.flower, "nonfinite_snapshot"):
            self._frames_unchecked = getattr(self, "_frames_unchecked", 0) + 1
            if synced or self._frames_unchecked >= every:
                self._frames_unchecked = 0
                self.flower.raise_if_nonfinite()
            return
        # No host wait (round 6).  Reading the counters with .item() every `every` frames drained the whole pipeline -- every queued
        # frame of both lanes: 24 ms each in the per-frame sharded mode, 40 % of its host time.  Instead, every `every` frames a
        # 16-byte snapshot of each engine's counter goes to pinned words behind the frame's kernels, and the PREVIOUS snapshot --
        # `every` frames old, long complete -- is what this call looks at: the raise comes at most 2 x `every` frames after the fact.
        # (A snapshot per frame was measured too: the per-frame sharded mode gains nothing more, and the host-io loop showed rare
        # 20-90 ms stalls in its result collection that it does not show without.)
        self._frames_unchecked = getattr(self, "_frames_unchecked", 0) + 1
        if self._frames_unchecked < every:
            return
        self._frames_unchecked = 0
        prev = getattr(self, "_nf_ring", None)
        if prev:
            ev, words = prev.pop(0)
            ev.synchronize()
            bad = int(words[:, 0].sum())
            if bad:
                self._nf_ring = []
                self.flower.nonfinite_count(reset=True)
                raise self.flower.nonfinite_error(bad)
        n = len(self.flower._all_engines())
        pool = getattr(self, "_nf_words", None)
        if pool is None or pool.shape[1] < n:
            pool = self._nf_words = torch.zeros((2, max(n, 4), 4), dtype=torch.int32).pin_memory()
            self._nf_slot = 0
        words = pool[self._nf_slot % 2][:n]
        self._nf_slot += 1
        self.flower.nonfinite_snapshot(words)
        self._nf_ring = [(torch.cuda.current_stream().record_event(), words)]

    def check_nonfinite(self):
        """Read the device-side non-finite counter now (synchronises) and raise FloatingPointError if it is not zero."""
        if hasattr(self.flower, "raise_if_nonfinite"):
            self._nf_ring = []
            self.flower.raise_if_nonfinite()

    def _flows_for_pairs(self, pairs, packed_out=None, planar=True):
        """[(left_id, left_img, right_id, right_img)] -> [(flow, occl, sigma[, packed])], one batched engine
        pass when the flow plugin offers one, else the reference's per-pair call (MFT/MFT.py:223-225).
        planar=False (native plugin only): just the packed results, (None, None, None, packed)."""
        if hasattr(self.flower, "compute_pairs"):
            if packed_out is not None and getattr(self.flower, "has_packed_output", False):
                return self.flower.compute_pairs(pairs, packed_out=packed_out, planar=planar)
            return self.flower.compute_pairs(pairs)
        res = []
        for _, left_img, _, right_img in pairs:
            f, extra = self.flower.compute_flow(left_img, right_img, mode='flow', init_flow=None)
            res.append((f, extra['occlusion'], extra['sigma']))
        return res

    def _flows_for(self, plan, right_id, input_img):
        """Right operands of the frame's chains -- FlowOU (left -> right_id) for every entry of ``plan``: cache
        first, then ONE batched flow computation for everything that is missing.  Each is either the planes
        tuple (flow, occl, sigma) or, fresh from the native engine, the packed [H, W, 4] tensor (planar copies
        are only produced when a cache wants them)."""
        out, missing = {}, []
        for i, (_, left_id, use_cache) in enumerate(plan):
            got = None
            if use_cache and self.flow_cache is not None:
                try:
                    f, o, s = self.flow_cache.read(left_id, right_id)
                    assert f is not None
                    got = FlowOUTrackingResult(f, o, s).planes()
                except Exception:
                    got = None
            if got is None:
                missing.append(i)
            else:
                # (a cache hit next to fresh packed results: packed once here, so that the selection stays on the packed
                # kernel instead of unpacking every fresh candidate)
                out[i] = pack_planes(got) if getattr(self.flower, "has_packed_output", False) else got
        if missing:
            to_cache = self.flow_cache is not None and any(plan[i][2] for i in missing)
            res = self._flows_for_pairs([(plan[i][1], self.memory[plan[i][1]]['img'], right_id, input_img)
                                         for i in missing], packed_out=True, planar=to_cache)
            for i, r in zip(missing, res):
                _, left_id, use_cache = plan[i]
                if self.flow_cache is not None and use_cache:
                    self.flow_cache.write(left_id, right_id, r[0], r[1], r[2])
                out[i] = r[3] if len(r) > 3 else tuple(r[:3])
        return [out[i] for i in range(len(plan))]

    # --------------------------------------------------------------- memory
    def cleanup_memory(self):
        """Keep the start frame (iff inf in deltas) and every frame still within
        the largest finite delta of the current one (MFT/MFT.py:157-181)."""
        finite = [d for d in self.C.deltas if np.isfinite(d)]
        max_delta = max(finite) if finite else 0
        has_direct_flow = any(np.isinf(d) for d in self.C.deltas)
        for mem_frame_i in list(self.memory.keys()):
            if mem_frame_i == self.start_frame_i and has_direct_flow:
                continue
            if self.time_direction > 0 and mem_frame_i + max_delta > self.current_frame_i:
                continue
            if self.time_direction < 0 and mem_frame_i - max_delta < self.current_frame_i:
                continue
            del self.memory[mem_frame_i]
        if hasattr(self.flower, "retain"):      # features of a look-ahead window's frames stay until they are tracked
            self.flower.retain(set(self.memory.keys()) | self._window_ids)

    def is_before_start(self, frame_i):
        return ((self.time_direction > 0 and frame_i < self.start_frame_i) or
                (self.time_direction < 0 and frame_i > self.start_frame_i))


def get_flowou_with_cache(flower, left_img, right_img, flow_init=None,
                          cache=None, left_id=None, right_id=None,
                          read_cache=False, write_cache=False):
    """Flow left -> right, possibly cached (MFT/MFT.py:189-230); any cache read
    error means recompute."""
    must_compute = not read_cache
    if read_cache and flow_init is None:
        assert left_id is not None
        assert right_id is not None
        try:
            assert cache is not None
            flow_left_to_right, occlusions, sigmas = cache.read(left_id, right_id)
            assert flow_left_to_right is not None
        except Exception:
            must_compute = True
    if must_compute:
        flow_left_to_right, extra = flower.compute_flow(left_img, right_img, mode='flow', init_flow=flow_init)
        occlusions, sigmas = extra['occlusion'], extra['sigma']
    if (cache is not None) and write_cache and must_compute and (flow_init is None):
        cache.write(left_id, right_id, flow_left_to_right, occlusions, sigmas)
    return FlowOUTrackingResult(flow_left_to_right, occlusions, sigmas)


def chain_results(left_result, right_result):
    """(template -> left) o (left -> right) in one HIP kernel (MFT/MFT.py:233-239)."""
    return left_result.chain_result(right_result)
