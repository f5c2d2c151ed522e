"""Flow-estimator plugin -- drop-in for ``MFT/raft.py:16-73`` (``RAFTWrapper``).

``config.flow_config.of_class(config.flow_config)`` must expose
``compute_flow(src_img, dst_img, mode='flow', init_flow=None) -> (flow,
{'occlusion', 'sigma', 'debug'})`` (``MFT/MFT.py:223-225``); this class does,
and adds ``compute_flow_many`` / ``compute_pairs``, which the tracker uses to run
all the flow deltas of a frame (or a share of a look-ahead window, ``dist.py``)
as ONE batched pass through the native engine.

Everything numeric runs in ``libmftx``: the feature /
context encoders (a3, ``mftx_encoder_forward``: each frame is encoded once and
cached -- fnet is per-sample instance-normalised and cnet uses eval-mode batch
norm, so per-frame encoding is exact) and the refinement a4-a12
(``mftx_raft_refine``).  PyTorch only owns the device buffers and the streams.
"""
from __future__ import annotations

import logging
import os
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .weights import make_weights, strip_module_prefix

logger = logging.getLogger(__name__)


def pad_amounts(H0, W0):
    """InputPadder('sintel') (core/utils/utils.py:9-19): (left, right, top, bottom)."""
    ph = (((H0 // 8) + 1) * 8 - H0) % 8
    pw = (((W0 // 8) + 1) * 8 - W0) % 8
    return pw // 2, pw - pw // 2, ph // 2, ph - ph // 2


# The plugin's own HIP streams (encoders / feature exchange, one per frame in flight) are created ONCE per process and device and
# shared by every plugin instance.  HIP hands its hardware queues to streams round robin as they are created: the first set of a
# process ends up on queues of its own (GPU_MAX_HW_QUEUES = 8, mft_amd/__init__.py), but the streams of a second, third, ... plugin
# instance may land on the queue of the caller's stream or of each other -- and a stream that waits for an event blocks the queue it
# shares: measured, trackers 2-4 of one process ran 153-160 frames/s where the first and a fresh process's run 174.  Instances that
# share a stream merely take turns on it.
_STREAMS = {}


def _shared_stream(device, kind, index=0):
    """(Instances -- also trackers in different threads -- that share a process share these streams: their work takes turns on
    them, in the order it was enqueued.)"""
    device = torch.device(device)
    dev_index = device.index if device.index is not None else torch.cuda.current_device()   # 'cuda' = the CURRENT device, not device 0
    key = (dev_index, kind, index)
    st = _STREAMS.get(key)
    if st is None:
        st = _STREAMS[key] = torch.cuda.Stream(device=torch.device("cuda", dev_index))
    return st


class FrameFeatures:
    __slots__ = ("fmap", "net", "inp", "h", "w", "pads", "shape", "ready")

    def __init__(self, fmap, net, inp, h, w, pads, shape, ready=None):
        self.fmap, self.net, self.inp, self.h, self.w, self.pads, self.shape = fmap, net, inp, h, w, pads, shape
        self.ready = ready          # event behind the kernels that wrote the maps (None: ordered by the caller's stream)


class RAFTWrapper:
    has_packed_output = True       # compute_pairs(packed_out=...) is supported
    STAGING_SLOTS = 4              # pinned staging buffers per frame shape for pageable host frames (_stage_host_frame)

    def __init__(self, config, device="cuda", state_dict=None):
        self.C = config
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("mft_amd.RAFTWrapper runs on an MI355X (HIP) device only; there is no CPU path")
        if state_dict is None:
            state_dict = self._load_weights(config)
        sd = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v)))
              for k, v in strip_module_prefix(state_dict).items()}
        self.sd = {k: v.to(self.device) for k, v in sd.items()}
        # raft_params.alternate_corr (core/raft.py:137-138): correlation on demand instead of the stored volume
        self._ondemand = bool(getattr(getattr(config, "raft_params", None), "alternate_corr", False))
        # raft_params.arith (this package's addition, default "split"): how the update block's matrix products are
        # formed -- "split": every fp32 operand as two fp16 halves on the fp16 matrix cores, fp32 accumulation (the
        # products carry ~2^-23 relative error, the same grade as an fp32 multiply), "fp32": fp32 MFMA
        arith = getattr(getattr(config, "raft_params", None), "arith", None) or "split"   # (a Config answers {} for a missing key)
        if arith not in ("split", "fp32"):
            raise ValueError(f"raft_params.arith must be 'split' or 'fp32', got {arith!r}")
        self._arith = ops.ARITH_SPLIT if arith == "split" else ops.ARITH_F32
        # The split arithmetic needs every operand below 65504 in magnitude (the fp16 range of the high halves).  Weights
        # are checked here, once (ops.split_weights): a checkpoint that does not fit runs on the fp32 matrix cores
        # instead, with the reason logged.  Activations are not clamped: one that leaves the range turns into NaN in
        # the outputs (never a finite wrong value); raft_params.check_finite = True checks every result and raises.
        self._engine_options = dict(getattr(getattr(config, "raft_params", None), "engine_options", None) or {})
        try:
            self._build_engines()
        except ops.SplitRangeError as e:
            logger.warning("raft_params.arith = 'split' refused (%s): falling back to arith = 'fp32'", e)
            self._arith = ops.ARITH_F32
            self._build_engines()
        self._check_finite = bool(getattr(getattr(config, "raft_params", None), "check_finite", False))
        # ... and whatever check_finite says, every refinement COUNTS its non-finite output pixels on the device (no host sync;
        # csrc/upsample.hip): raise_if_nonfinite() is called by the tracker whenever it synchronises anyway
        # C.split_streams (env MFTX_SPLIT_STREAMS overrides): batches of >= 6 pairs run as that many parts on
        # separate HIP streams (see _refine_split); 1 = one stream.  Measured at 7 pairs, 512 x 512
        # (profiles/r2_split_streams.txt): 1 / 2 / 3 / 4 / 7 parts = 63.0 / 64.9 / 59.4 / 60.6 / 49.8 frames/s with fp32 MFMA
        # (64 x 64 tiles: a half batch still fills the chip); with the split arithmetic the big tiles want the whole
        # batch: 1 / 2 parts = 104.9 / 100.6 frames/s.  Default: 1 (split), 2 (fp32).
        self._split_streams = int(os.environ.get("MFTX_SPLIT_STREAMS", "") or getattr(config, "split_streams", 0) or
                                  (1 if self._arith == ops.ARITH_SPLIT else 2))
        self._engines, self._side = [], []            # part k: engine (own workspace) and stream (None = caller's)
        # C.frames_in_flight (env MFTX_FRAMES_IN_FLIGHT overrides; default 1): consecutive compute_pairs calls -- the tracker's
        # frames -- alternate between that many LANES, each an engine with its own workspace on its own HIP stream.  The flow
        # batch of frame t + 1 depends on frame t's FEATURES only (chaining and selection, which need frame t's result, stay on
        # the caller's stream behind an event), so with the host ahead of the GPU two frames' kernel chains are in flight at
        # once and the hardware interleaves their workgroups: a 224-workgroup GEMM kernel of one frame leaves 32 CUs and its
        # load / epilogue phases to the other frame's kernels, and a frame's store-bound volume kernel, OU heads and upsampling
        # run beside the other frame's matrix work instead of alone.  Same kernels, same batches: the same bits.
        # Measured (512 x 512, 7 pairs, one MI355X, same box): 1 / 2 / 3 lanes = 162.4 / 178.0 / 165.2 frames/s.  Default: 2 with the
        # split arithmetic (the fp32 path batches through _refine_split instead), 1 otherwise.
        self._fif = self._frames_in_flight_setting(config)
        self._lanes, self._lane_next = [], 0          # [(engine, stream)]
        self._lanes_stale = False
        # ... and the host may not run ahead of the GPU without bound (every queued batch holds its outputs, allocated when it is
        # enqueued: 29 MB per 512 x 512 frame of 7 pairs): at most C.max_batches_ahead (default 16) lane batches are pending, the
        # call that would queue one more first waits for the oldest.  Far above what keeps two lanes busy, never reached by a caller
        # that consumes results as they complete.
        self._ahead_max = max(2, int(getattr(config, "max_batches_ahead", 0) or 16))
        self._ahead = []                              # done events of the queued lane batches, oldest first
        # Which GEMM kernels a refinement runs on (tile-resident or ring-buffered: they differ by fp32 rounding of the K sums) must
        # not depend on the batch a pair happens to ride in -- a tracker's ramp-up frames, a remainder window or one rank's share
        # of a sharded job would then give other bits than the full batch.  The choice is made ONCE per image size, for the
        # NOMINAL batch (the tracker's delta count; MFT.init sets it), and pinned on every engine of this plugin
        # (mftx_tile_conv_fills_chip; an explicit engine_options["tile_conv"] wins).
        self.nominal_pairs = int(getattr(config, "nominal_pairs", 0) or 7)
        self._tile_choice = {}
        self._frames = {}
        self._staging = {}                            # frame shape -> pinned staging ring (_stage_host_frame)
        # Optional (C.async_encode): encode new frames on a side stream.  The encoders of frame t
        # only need the image, so with results kept on the device (no per-frame host sync) their
        # kernels overlap the tail of frame t-1's GEMM-bound refinement instead of sitting on the
        # critical path.  Device-tensor frames must be complete when passed in.
        # Default (the key absent from the config): ON when several frames are in flight (C.frames_in_flight below) -- a lane waits
        # for the encoders' event only, and encoders queued on the caller's stream would sit behind the previous frame's selection,
        # i.e. behind the previous frame's whole batch; `async_encode = False` said explicitly still wins.
        ae = getattr(config, "__dict__", {}).get("async_encode", None)
        # (on by DEFAULT it is also safe for frames that are device tensors still being produced on the caller's stream: such a frame
        # is encoded on the caller's stream, in order.  `async_encode = True` said explicitly keeps the contract above -- device frames
        # complete when passed in -- and the overlap that goes with it.  Host frames are uploaded on the encode stream either way.)
        self._enc_waits_for_device_frames = ae is None
        if ae is None:
            ae = self._frames_in_flight_setting(config) > 1
        self._enc_stream = _shared_stream(self.device, "enc") if ae else None

    def _frames_in_flight_setting(self, config):
        return max(1, int(os.environ.get("MFTX_FRAMES_IN_FLIGHT", "") or getattr(config, "frames_in_flight", 0) or
                          (2 if self._arith == ops.ARITH_SPLIT else 1)))

    def _build_engines(self):
        graph = bool(self._engine_options.get("graph", 1))
        self.fnet_engine = ops.EncoderEngine(self.sd, "fnet", True, self.device, arith=self._arith, graph=graph)
        self.cnet_engine = ops.EncoderEngine(self.sd, "cnet", False, self.device, arith=self._arith, graph=graph)
        self.engine = ops.RaftEngine(self.sd, self.device, ondemand_corr=self._ondemand, arith=self._arith,
                                     options=self._engine_options)

    def set_nominal_pairs(self, n):
        """The batch size kernel choices are made for (see __init__); changing it re-decides at the next call."""
        n = max(1, int(n))
        if n != self.nominal_pairs:
            self.nominal_pairs = n
            self._tile_choice = {}

    def _pin_kernels(self, h, w):
        if self._arith != ops.ARITH_SPLIT or "tile_conv" in self._engine_options:
            return
        v = self._tile_choice.get((h, w))
        if v is None:
            v = self._tile_choice[(h, w)] = 2 if ops._lib.load().mftx_tile_conv_fills_chip(self.nominal_pairs, h, w) else 0
        for e in self._all_engines():
            if getattr(e, "_tile_conv", None) != v:
                e.set_option("tile_conv", v)

    def _all_engines(self):
        return list(dict.fromkeys([self.engine] + list(self._engines) + [e for e, _ in self._lanes]))

    def _lane(self):
        """The next lane (engine, stream) of C.frames_in_flight; lane 0 is the plugin's first engine."""
        while len(self._lanes) < self._fif:
            eng = self.engine if not self._lanes else ops.RaftEngine(self.sd, self.device, ondemand_corr=self._ondemand,
                                                                      arith=self._arith, options=self._engine_options)
            self._lanes.append((eng, _shared_stream(self.device, "lane", len(self._lanes))))      # (stream priorities: measured, no effect)
            self._lanes_stale = True              # (a new engine packs its weights on the caller's stream: the lanes wait for that)
        lane = self._lanes[self._lane_next % self._fif]
        self._lane_next += 1
        return lane

    def _lanes_fit(self, P, h, w):
        """Every lane beyond the first owns a workspace of its own (1.1 GB at 7 pairs of 512 x 512, 43 GB at 7 pairs of 1080p): before
        a lane allocates one, check that it fits into what the device has free NEXT TO what is already allocated (a flow cache's HBM
        tier, the caller's own tensors) with a tenth of the device kept in reserve -- otherwise run with the lanes that exist
        (frames_in_flight is lowered for this plugin, logged once) instead of failing in the middle of a sequence."""
        while self._fif > max(1, len(self._lanes)):
            k = max(1, len(self._lanes))              # the lane the next _lane() call would create first (lane 0 is the plugin's first
            need = int(ops._lib.load().mftx_raft_workspace_bytes_for(self.engine._h, P, h, w))   # engine: needed whatever the setting)
            free, total = torch.cuda.mem_get_info(self.device)
            cached = torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device)   # (torch's pool can be re-used)
            if need <= free + cached - total // 10:
                return
            logger.warning("frames_in_flight = %d needs %.1f GB more workspace (lane %d, %d pairs of %d x %d cells) but only %.1f GB of "
                           "the device's %.1f GB are free: running with %d frame(s) in flight", self._fif, need / 1e9, k, P, h, w,
                           (free + cached) / 1e9, total / 1e9, k)
            self._fif = k

    def nonfinite_count(self, reset=False):
        """Non-finite output pixels counted on the device since the last reset, over all engines of this plugin."""
        return sum(e.nonfinite_count(reset=reset) for e in self._all_engines())

    def nonfinite_snapshot(self, host_words):
        """Asynchronous read of the counters (no host wait): enqueues, on the current stream, copies of every engine's counter into
        ``host_words`` -- pinned int32 [n, 4] with n >= the number of engines; the caller sums column 0 after waiting for an
        event recorded behind this call (mft_amd.video.ResultDrain does)."""
        for i, e in enumerate(self._all_engines()):
            e.nonfinite_snapshot(host_words[i])

    def nonfinite_error(self, bad):
        return FloatingPointError(
            f"flow network: {bad} output pixels with a non-finite value (a NaN, or an infinite flow / occlusion)" + (
                " -- an activation left the fp16 range of the split arithmetic (|x| >= 65504); "
                "set raft_params.arith = 'fp32'" if self._arith == ops.ARITH_SPLIT else ""))

    def raise_if_nonfinite(self):
        bad = self.nonfinite_count(reset=True)
        if bad:
            raise self.nonfinite_error(bad)

    @property
    def arith(self):
        """'split' or 'fp32': the arithmetic in use (a checkpoint outside the fp16 range falls back to 'fp32')."""
        return "split" if self._arith == ops.ARITH_SPLIT else "fp32"

    @staticmethod
    def _load_weights(config):
        """``torch.load(C.model)`` like the reference (MFT/raft.py:20-21).  A missing checkpoint is an
        error, as there; seeded synthetic weights (benchmarks and tests: the trained checkpoint is not
        distributed with this build) must be asked for explicitly with ``C.model = None`` and an integer
        ``C.synthetic_weights_seed``."""
        model = getattr(config, "model", None)
        seed = getattr(config, "synthetic_weights_seed", None)
        if model:
            if not Path(str(model)).exists():
                raise FileNotFoundError(
                    f"flow checkpoint {model!r} not found (set flow_config.model = None and "
                    "flow_config.synthetic_weights_seed = <int> to run on seeded synthetic weights)")
            logger.info("loading checkpoint %s", model)
            return torch.load(model, map_location="cpu")
        if not isinstance(seed, int) or isinstance(seed, bool):
            raise ValueError("no flow checkpoint configured: set flow_config.model, or "
                             "flow_config.synthetic_weights_seed = <int> for seeded synthetic weights")
        logger.warning("no checkpoint configured: using seeded synthetic weights (seed %d)", seed)
        return make_weights(seed)

    # ---- per-frame encoding + cache ---------------------------------------
    @staticmethod
    def _geometry(H0, W0):
        pads = pad_amounts(H0, W0)
        return (H0 + pads[2] + pads[3]) // 8, (W0 + pads[0] + pads[1]) // 8, pads

    def _device_image(self, img_bgr):
        img = img_bgr if isinstance(img_bgr, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(img_bgr))
        if img.is_cuda:
            return img.to(self.device, non_blocking=True).contiguous()
        if img.dtype == torch.uint8 and not (img.is_pinned() and img.is_contiguous() and img.data_ptr() % 16 == 0):
            # a plain (pageable) host frame -- what the reference's loop hands over (numpy from cv2, MFT/utils/io.py:566-615): staged
            # through a small rotation of pinned buffers owned by the plugin (a plain memcpy, ~0.1 ms for 786 kB) instead of a
            # synchronous pageable hipMemcpy, so that the upload is asynchronous like FrameRing's.  The caller may recycle its
            # array as soon as this returns.
            img = self._stage_host_frame(img)
        if img.is_pinned() and img.is_contiguous() and img.dtype == torch.uint8 and img.data_ptr() % 16 == 0:
            # a pinned host frame (mft_amd.video.FrameRing or the staging above): uploaded by a copy KERNEL (16-byte coalesced
            # reads over PCIe) on the current stream instead of hipMemcpyAsync -- no SDMA queue, nothing to serialise behind a
            # pending download
            dev = torch.empty(img.shape, dtype=torch.uint8, device=self.device)
            ops.copy_bytes(img, dev)
            st = getattr(img, "_mftx_slot", None)
            if st is not None:                    # the staging slot is free again once this upload has run
                st[1] = torch.cuda.current_stream(self.device).record_event()
            return dev
        return img.to(self.device, non_blocking=True).contiguous()

    def _stage_host_frame(self, img):
        """Pageable uint8 frame -> the next of the plugin's pinned staging buffers of that shape (allocated on first use, pinning is
        slow).  A buffer is rewritten only after the upload that read it last has completed (its event; in steady state that was
        STAGING_SLOTS frames ago and costs nothing)."""
        key = tuple(img.shape)
        ring = self._staging.get(key)
        if ring is None:
            if len(self._staging) >= 4:           # a plugin fed many different sizes: keep the pinned footprint bounded
                self._staging.pop(next(iter(self._staging)))
            ring = self._staging[key] = {"n": 0, "slots": []}
        if len(ring["slots"]) < self.STAGING_SLOTS:
            buf = torch.empty(key, dtype=torch.uint8).pin_memory()
            slot = [buf, None]
            buf._mftx_slot = slot
            ring["slots"].append(slot)
        else:
            slot = ring["slots"][ring["n"] % self.STAGING_SLOTS]
        ring["n"] += 1
        if slot[1] is not None:
            slot[1].synchronize()
            slot[1] = None
        # (numpy: a plain single-threaded memcpy; torch's CPU copy_ would wake the whole intra-op pool for 786 kB, mft_amd/video.py)
        np.copyto(slot[0].numpy(), img.numpy() if img.is_contiguous() else np.ascontiguousarray(img.numpy()))
        return slot[0]

    @torch.no_grad()
    def encode(self, img_bgr, want_context=True) -> FrameFeatures:
        """uint8 BGR (H,W,3) -> pixel-major features (MFT/raft.py:41-48, core/raft.py:122-149).
        (fnet and cnet are independent, and running cnet on a second stream beside fnet was measured in round 6: 183 -> 162
        frames/s pipelined and 143 -> 117 with a host synchronisation per frame -- two chains of ~50 small launches side by side
        cost more than they hide.  One after the other, on one stream.)"""
        H0, W0 = img_bgr.shape[:2]
        img = self._device_image(img_bgr)
        h, w, pads = self._geometry(H0, W0)
        fmap, _ = self.fnet_engine.forward(img)
        net = inp = None
        if want_context:
            net, inp = self.cnet_engine.forward(img)
        return FrameFeatures(fmap, net, inp, h, w, pads, (H0, W0))

    def ensure_encode_stream(self):
        """The stream every encoder launch of this plugin runs on from now on (the encoder engines own ONE workspace
        each: encodes issued from different streams would race on it)."""
        if self._enc_stream is None:
            self._enc_stream = _shared_stream(self.device, "enc")
            self._enc_stream.wait_stream(torch.cuda.current_stream(self.device))   # encodes already queued elsewhere finish first
        return self._enc_stream

    @torch.no_grad()
    def encode_packed(self, img_bgr, wait=True):
        """All features of a frame in ONE buffer [h*w*512] = fmap [N,256] | net [N,128] | inp [N,128]
        (each block contiguous), the unit the multi-GPU path all-gathers.  -> (buffer, (h, w)).
        Runs on the encode stream when there is one; wait=False leaves the caller's stream un-synchronised with it
        (the caller orders later consumers itself, e.g. behind a collective issued on the encode stream)."""
        H0, W0 = img_bgr.shape[:2]
        h, w, _ = self._geometry(H0, W0)
        N = h * w

        def run():
            img = self._device_image(img_bgr)
            buf = torch.empty(N * 512, dtype=torch.float32, device=self.device)
            self.fnet_engine.forward(img, out=(buf[: N * 256].view(N, 256), None))
            self.cnet_engine.forward(img, out=(buf[N * 256: N * 384].view(N, 128), buf[N * 384:].view(N, 128)))
            return buf

        if self._enc_stream is None or torch.cuda.current_stream(self.device) == self._enc_stream:
            cur = torch.cuda.current_stream(self.device)
            self._encode_begin(cur)
            buf = run()
            self._encode_end(cur.record_event(), cur)
            return buf, (h, w)
        main = torch.cuda.current_stream(self.device)
        self._encode_begin(self._enc_stream)
        with torch.cuda.stream(self._enc_stream):
            buf = run()
            self._encode_end(self._enc_stream.record_event(), self._enc_stream)
        if wait:
            main.wait_stream(self._enc_stream)
            buf.record_stream(main)
        return buf, (h, w)

    @torch.no_grad()
    def encode_half(self, img_bgr, part, wait=True):
        """One of the two encoders of a frame: part 0 = fnet -> fmap [N, 256], part 1 = cnet -> net [N, 128] | inp [N, 128]; both
        are N * 256 floats -- the two halves of an ``encode_packed`` buffer.  The multi-GPU path hands the two networks of a frame
        to two ranks when a window has fewer frames than half the ranks (the per-frame mode): the serial head of the frame is one
        encoder instead of two.  Stream handling as in ``encode_packed``."""
        H0, W0 = img_bgr.shape[:2]
        h, w, _ = self._geometry(H0, W0)
        N = h * w

        def run():
            img = self._device_image(img_bgr)
            buf = torch.empty(N * 256, dtype=torch.float32, device=self.device)
            if part == 0:
                self.fnet_engine.forward(img, out=(buf.view(N, 256), None))
            else:
                self.cnet_engine.forward(img, out=(buf[: N * 128].view(N, 128), buf[N * 128:].view(N, 128)))
            return buf

        if self._enc_stream is None or torch.cuda.current_stream(self.device) == self._enc_stream:
            cur = torch.cuda.current_stream(self.device)
            self._encode_begin(cur)
            buf = run()
            self._encode_end(cur.record_event(), cur)
            return buf, (h, w)
        main = torch.cuda.current_stream(self.device)
        self._encode_begin(self._enc_stream)
        with torch.cuda.stream(self._enc_stream):
            buf = run()
            self._encode_end(self._enc_stream.record_event(), self._enc_stream)
        if wait:
            main.wait_stream(self._enc_stream)
            buf.record_stream(main)
        return buf, (h, w)

    def adopt_halves(self, frame_id, fbuf, cbuf, img_bgr, ready=None):
        """Install a frame's features from the two ``encode_half`` buffers (produced here or on other ranks)."""
        H0, W0 = img_bgr.shape[:2]
        h, w, pads = self._geometry(H0, W0)
        N = h * w
        assert fbuf.numel() == N * 256 and cbuf.numel() == N * 256
        fbuf, cbuf = fbuf.reshape(-1), cbuf.reshape(-1)
        self._frames[frame_id] = FrameFeatures(fbuf.view(N, 256), cbuf[: N * 128].view(N, 128), cbuf[N * 128:].view(N, 128),
                                               h, w, pads, (H0, W0), ready=ready)

    def packed_numel(self, img_bgr):
        """Floats in the buffer ``encode_packed`` produces for a frame of this size."""
        h, w, _ = self._geometry(*img_bgr.shape[:2])
        return h * w * 512

    def adopt_packed(self, frame_id, buf, img_bgr, ready=None):
        """Install features produced by ``encode_packed`` (here or on another rank) for ``frame_id``.  ``ready``: an event behind
        whatever wrote ``buf`` (None: ordered by the stream that is current when the features are used)."""
        H0, W0 = img_bgr.shape[:2]
        h, w, pads = self._geometry(H0, W0)
        N = h * w
        assert buf.numel() == N * 512
        buf = buf.reshape(-1)
        self._frames[frame_id] = FrameFeatures(buf[: N * 256].view(N, 256), buf[N * 256: N * 384].view(N, 128),
                                               buf[N * 384:].view(N, 128), h, w, pads, (H0, W0), ready=ready)

    def reset_cache(self):
        self._frames = {}

    def retain(self, frame_ids):
        keep = set(frame_ids)
        for k in [k for k in self._frames if k not in keep]:
            del self._frames[k]

    def _encode(self, img):
        # Every set of features carries the event behind its encoders, whatever frames_in_flight says right now: features live in the
        # cache for up to 32 frames, and a lane that meets one WITHOUT an event must wait for the caller's whole stream -- i.e. for the
        # previous frame's selection -- which silently serialises the lanes (round 5: a plugin whose frames_in_flight was lowered
        # and raised again ran at the one-lane rate until the last event-less frame had left the cache: bench.py's host-io pass
        # behind its profile pass, 152 instead of 176 frames/s).
        #   The encoder engines own ONE workspace each: an encode that runs on another stream than the previous one waits for it
        # (_encode_begin / _encode_end: the default-async branch below, a plugin whose encode stream is switched on a live tracker --
        # bench.py's profile pass does --, the sharded path's prefetch beside caller-stream encodes).
        if self._enc_stream is None:
            cur = torch.cuda.current_stream()
            self._encode_begin(cur)
            f = self.encode(img)
            f.ready = cur.record_event()
            self._encode_end(f.ready, cur)
            return f
        main = torch.cuda.current_stream()
        if self._enc_waits_for_device_frames and isinstance(img, torch.Tensor) and img.is_cuda:
            # async_encode by DEFAULT and a frame that is a device tensor: it may still be being written on the caller's stream, so it
            # is encoded there, in order (as before round 5; the encode stream waiting for the caller's stream would cost more: 142
            # against 160 frames/s).  Frames known to be complete: say async_encode = True.
            # The encoder engines own ONE workspace each, and other encodes of this plugin (host frames, the sharded path's
            # prefetch: encode_packed / encode_half) run on the encode stream: this encode waits for those, and they for it.
            self._encode_begin(main)
            f = self.encode(img)
            f.ready = main.record_event()
            self._encode_end(f.ready, main)
            return f
        self._encode_begin(self._enc_stream)
        with torch.cuda.stream(self._enc_stream):
            f = self.encode(img)
            f.ready = self._enc_stream.record_event()
        self._encode_end(f.ready, self._enc_stream)
        main.wait_stream(self._enc_stream)
        for t in (f.fmap, f.net, f.inp):          # allocated on the side stream, consumed on `main`
            if t is not None:
                t.record_stream(main)
        return f

    def _encode_begin(self, stream):
        prev = getattr(self, "_last_encode", None)
        if prev is not None and prev[1] != stream:
            stream.wait_event(prev[0])

    def _encode_end(self, event, stream):
        self._last_encode = (event, stream)

    def _features(self, key, img):
        if key is None:
            return self._encode(img)
        f = self._frames.get(key)
        if f is None or f.shape != img.shape[:2]:
            f = self._encode(img)
            self._frames[key] = f
        return f

    # ---- batched entry points used by the tracker --------------------------
    @torch.no_grad()
    def compute_pairs(self, pairs, iters=None, init_flow=None, packed_out=None, planar=True):
        """pairs: [(left_id | None, left_img, right_id | None, right_img)] -> [(flow[2,H,W],
        occl[1,H,W], sigma[1,H,W])], left_i -> right_i for every i, in ONE engine call.  Frame ids key
        the feature cache (None = do not cache).  init_flow: optional [P,2,H,W] initial flows.
        packed_out: True, or a pre-allocated [P,H,W,4] tensor: the tuples get a fourth element, the same result
        interleaved per pixel (fx, fy, occl, sigma) -- what ``mftx_chain_select_packed`` gathers from;
        planar=False then skips the planar outputs (the first three elements are None)."""
        iters = int(iters if iters is not None else self.C.flow_iters)
        fls = [self._features(lk, li) for lk, li, _, _ in pairs]
        frs = [self._features(rk, ri) for _, _, rk, ri in pairs]
        ref = fls[0]
        for f in fls + frs:
            if f.shape != ref.shape:
                raise ValueError("all frames of a batch must have the same size")
        P = len(pairs)
        gather = self.engine.can_gather(P) and not (self._split_streams > 1 and P >= 6 and init_flow is None)
        if gather:      # the engine takes the pairs' cached maps where they lie: no batch tensors, a shared right frame split once
            fmap1, fmap2 = [f.fmap for f in fls], [f.fmap for f in frs]
            net, inp = [f.net for f in fls], [f.inp for f in fls]
        else:
            fmap1 = torch.stack([f.fmap for f in fls])
            fmap2 = torch.stack([f.fmap for f in frs])
            net = torch.stack([f.net for f in fls])
            inp = torch.stack([f.inp for f in fls])
        flow_init = None
        if init_flow is not None:
            flow_init = self._init_flow_lr(init_flow, ref)
        packed = None
        H0, W0 = ref.shape
        want_planar = planar or packed_out is None or packed_out is False
        if (self._fif > 1 and gather and flow_init is None and not self._check_finite
                and (packed_out is None or isinstance(packed_out, bool))):
            # (one frame's batch as two halves on the two lanes -- for the caller that synchronises per frame -- was measured in
            # round 6: the pipelined rate falls 188 -> 171 frames/s and the per-frame-synchronised rate does not move, 150 -> 150.)
            return self._refine_on_lane(fls, frs, fmap1, fmap2, net, inp, ref, iters, bool(packed_out), want_planar)
        if packed_out is not None and packed_out is not False:
            packed = packed_out if isinstance(packed_out, torch.Tensor) else \
                torch.empty(P, H0, W0, 4, dtype=torch.float32, device=self.device)
        self._pin_kernels(ref.h, ref.w)
        if self._lanes:                               # (lane 0 shares this engine's workspace: in-flight lane work finishes first)
            for _, s_ in self._lanes:
                torch.cuda.current_stream(self.device).wait_stream(s_)
            self._lanes_stale = True
        if self._split_streams > 1 and P >= 6 and flow_init is None:
            flow, occl, sigma = self._refine_split(fmap1, fmap2, net, inp, ref, iters, packed, want_planar)
        else:
            flow, occl, sigma = self.engine.refine(fmap1, fmap2, net, inp, ref.h, ref.w, iters, pads=ref.pads,
                                                   flow_init=flow_init, packed=packed, planar=want_planar)
        if self._check_finite:
            bad = sum(ops.count_not_below(t.reshape(-1), float("inf")) for t in (packed, flow, occl, sigma) if t is not None)
            if bad:
                raise FloatingPointError(
                    f"compute_flow: {bad} non-finite output values" + (
                        " -- an activation left the fp16 range of the split arithmetic (|x| >= 65504); "
                        "set raft_params.arith = 'fp32'" if self._arith == ops.ARITH_SPLIT else ""))
        if packed is not None and flow is None:
            return [(None, None, None, packed[i]) for i in range(P)]
        if packed is not None:
            return [(flow[i], occl[i], sigma[i], packed[i]) for i in range(P)]
        return [(flow[i], occl[i], sigma[i]) for i in range(P)]

    def _refine_on_lane(self, fls, frs, fmap1, fmap2, net, inp, geom, iters, want_packed, planar):
        """One batch on the next lane of C.frames_in_flight (see __init__): the lane's stream waits for the frames' features
        only, the caller's stream for the lane -- whatever else is queued on the caller's stream (the previous frame's chaining
        and selection, result copies) does not hold the batch back."""
        P = len(fls)
        H0, W0 = geom.shape
        self._lanes_fit(P, geom.h, geom.w)
        eng, st = self._lane()
        main = torch.cuda.current_stream(self.device)
        self._pin_kernels(geom.h, geom.w)
        if self._lanes_stale:                         # an engine ran on the caller's stream since the lanes last did: order them behind it
            for _, s_ in self._lanes:
                s_.wait_stream(main)
            self._lanes_stale = False
        for f in dict.fromkeys(fls + frs):
            if f.ready is not None:
                st.wait_event(f.ready)
            else:
                st.wait_stream(main)                  # features without an event: ordered by the caller's stream
            for t in (f.fmap, f.net, f.inp):
                if t is not None:
                    t.record_stream(st)
        with torch.cuda.stream(st):
            # (outputs come from the LANE's pool: a block the caller's stream freed a moment ago may still be read there)
            packed = torch.empty(P, H0, W0, 4, dtype=torch.float32, device=self.device) if want_packed else None
            flow, occl, sigma = eng.refine(fmap1, fmap2, net, inp, geom.h, geom.w, iters, pads=geom.pads,
                                           flow_init=None, packed=packed, planar=planar)
            done = st.record_event()
        main.wait_event(done)
        self._ahead.append(done)
        while len(self._ahead) > self._ahead_max:
            self._ahead.pop(0).synchronize()
        for t in (packed, flow, occl, sigma):
            if t is not None:
                t.record_stream(main)
        if packed is not None and flow is None:
            return [(None, None, None, packed[i]) for i in range(P)]
        if packed is not None:
            return [(flow[i], occl[i], sigma[i], packed[i]) for i in range(P)]
        return [(flow[i], occl[i], sigma[i]) for i in range(P)]

    def _refine_split(self, fmap1, fmap2, net, inp, geom, iters, packed, planar):
        """The batch as ``C.split_streams`` parts on as many HIP streams, each with its own workspace: one part's
        kernel tails, launch gaps and ragged last tile rounds are filled by the other parts' kernels (the
        hardware interleaves workgroups of kernels from different streams).  Same per-pair results -- the
        kernels are batch-invariant (``test_batch_invariance_bitwise``)."""
        P = fmap1.shape[0]
        H0, W0 = geom.shape
        S = min(self._split_streams, P)
        while len(self._engines) < S:
            self._engines.append(ops.RaftEngine(self.sd, self.device, ondemand_corr=self._ondemand, arith=self._arith,
                                                options=self._engine_options)
                                 if self._engines else self.engine)
            # part 0 runs on the calling stream, the others on side streams (HIP multiplexes streams onto a handful
            # of hardware queues: every stream saved keeps the copy / encoder streams on queues of their own)
            self._side.append(torch.cuda.Stream(device=self.device) if self._side or len(self._engines) > 1 else None)
        self._pin_kernels(geom.h, geom.w)             # the parts run the kernels the WHOLE batch would
        dev = self.device
        flow = torch.empty(P, 2, H0, W0, dtype=torch.float32, device=dev) if planar else None
        occl = torch.empty(P, 1, H0, W0, dtype=torch.float32, device=dev) if planar else None
        sigma = torch.empty(P, 1, H0, W0, dtype=torch.float32, device=dev) if planar else None
        bounds = [(P * k) // S for k in range(S + 1)]
        parts = [(self._engines[k], self._side[k], slice(bounds[k], bounds[k + 1])) for k in range(S)]
        for eng, _, sl in parts:                      # workspaces are allocated up front, on the calling stream
            eng.workspace(sl.stop - sl.start, geom.h, geom.w)
        main = torch.cuda.current_stream()

        def run(eng, sl):
            f, o, s_ = eng.refine(fmap1[sl], fmap2[sl], net[sl], inp[sl], geom.h, geom.w, iters, pads=geom.pads,
                                  packed=packed[sl] if packed is not None else None, planar=planar)
            if planar:
                flow[sl].copy_(f); occl[sl].copy_(o); sigma[sl].copy_(s_)

        for eng, st, sl in parts[1:]:                 # side streams first: they start as soon as the inputs exist
            st.wait_stream(main)
            with torch.cuda.stream(st):
                run(eng, sl)
        run(parts[0][0], parts[0][2])
        for _, st, _ in parts[1:]:
            main.wait_stream(st)
        return flow, occl, sigma

    def compute_flow_many(self, lefts, right, iters=None):
        """lefts: [(frame_id | None, img)], right: (frame_id | None, img): left_i -> right for every i."""
        return self.compute_pairs([(k, im, right[0], right[1]) for k, im in lefts], iters)

    def _init_flow_lr(self, init_flow, geom):
        """[P,2,H0,W0] full-resolution initial flow -> pixel-major [P, h*w, 2] at 1/8 resolution, the way
        the reference prepares ``flow_init`` (MFT/raft.py:49-52,98-101): replicate-pad like the images,
        bilinear resize with align_corners=True, divide by 8.  (Host-side plumbing of an argument MFT
        itself never passes, MFT/MFT.py:98.)"""
        x = torch.as_tensor(init_flow).to(self.device, torch.float32)
        if x.dim() == 3:
            x = x[None]
        x = F.pad(x, list(geom.pads), mode="replicate")
        x = F.interpolate(x, size=(geom.h, geom.w), mode="bilinear", align_corners=True) / 8
        return x.permute(0, 2, 3, 1).reshape(x.shape[0], geom.h * geom.w, 2).contiguous()

    # ---- reference plugin API ---------------------------------------------
    @torch.no_grad()
    def compute_flow(self, src_img, dst_img, mode="TC", vis=False, src_img_identifier=None, numpy_out=False,
                     init_flow=None, vis_debug=False):
        """(H,W,3) uint8 BGR images -> flow (2,H,W) + {'occlusion','sigma','debug'}
        (mode='flow'), or (src_coords, dst_coords, extra) (mode='TC')."""
        H, W = src_img.shape[:2]
        debug = None
        if vis_debug:
            # the reference's debug payload (core/raft.py:159-176, 255-257): cost-volume pyramid, start grid and the
            # coordinates of every iteration, on the CPU
            fl, fr = self._features(None, src_img), self._features(None, dst_img)
            flow_init = self._init_flow_lr(init_flow, fl) if init_flow is not None else None
            self._pin_kernels(fl.h, fl.w)
            if self._lanes:
                for _, s_ in self._lanes:
                    torch.cuda.current_stream(self.device).wait_stream(s_)
                self._lanes_stale = True
            (flow, occl, sigma), debug = self.engine.debug_refine(fl.fmap[None], fr.fmap[None], fl.net[None], fl.inp[None],
                                                                   fl.h, fl.w, int(self.C.flow_iters), pads=fl.pads,
                                                                   flow_init=flow_init)
            flow, occl, sigma = flow[0], occl[0], sigma[0]
        else:
            (flow, occl, sigma), = self.compute_pairs([(None, src_img, None, dst_img)], init_flow=init_flow)
        assert flow.shape == (2, H, W)
        conv = (lambda t: t.detach().cpu().numpy()) if numpy_out else (lambda t: t)
        if mode == "flow":
            return conv(flow), {"occlusion": conv(occl), "sigma": conv(sigma), "debug": debug}
        if mode == "TC":
            idx = torch.arange(H * W, device=flow.device)
            src = torch.stack([idx % W, torch.div(idx, W, rounding_mode="floor")]).to(torch.float32)
            dst = src + flow.reshape(2, H * W)
            return conv(src), conv(dst), {"occlusion": conv(occl.reshape(-1)) if numpy_out else occl,
                                          "sigma": conv(sigma.reshape(-1)) if numpy_out else sigma,
                                          "debug": debug}
        raise ValueError(f"unknown mode {mode!r}")
