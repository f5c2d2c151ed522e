"""Video ingest -- drop-in for ``GeneralVideoCapture`` / ``get_video_frames`` / ``get_video_length``
(``MFT/utils/io.py:566-615``) plus what the MI355X path adds: frames staged in pinned host memory so that their upload is asynchronous and
overlaps the previous frame's refinement, and the mirror for results coming back.

Sources: a directory of images (``.png`` decoded by this package's own PNG coder; ``.jpg`` / ``.jpeg`` and
container formats need OpenCV, which is used when it is importable and reported clearly when it is not), or
a ``.npy`` / ``.npz`` array of frames ``[T, H, W, 3]`` uint8 BGR.  Frames are BGR ``(H, W, 3)`` uint8 like
``cv2.imread`` returns them.
"""
from __future__ import annotations

import os
from pathlib import Path

import numpy as np
import torch

IMAGE_SUFFIXES = ('.jpg', '.png', '.jpeg')

def _cv2():
    try:
        import cv2
        return cv2
    except ImportError:
        return None


def imread_bgr(path):
    """``cv2.imread(path)``: 8-bit BGR (alpha dropped, 16-bit scaled down) -- PNG natively, the rest through cv2."""
    path = Path(path)
    if path.suffix.lower() == '.png':
        from .flowou_codec import png_decode
        img = png_decode(path.read_bytes())
        if img.dtype == np.uint16:
            img = (img >> 8).astype(np.uint8)
        return np.ascontiguousarray(img[..., [2, 1, 0]])
    cv2 = _cv2()
    if cv2 is None:
        raise RuntimeError(f"cannot decode {path.name}: only PNG is decoded natively, {path.suffix} needs OpenCV (cv2)")
    return cv2.imread(str(path))


def imread_unchanged(path):
    """``cv2.imread(path, cv2.IMREAD_UNCHANGED)`` for a PNG: (B, G, R[, A]) channel order, depth kept."""
    from .flowou_codec import cv2_imdecode_png
    return cv2_imdecode_png(Path(path).read_bytes())


def imwrite_bgr(path, img):
    """``cv2.imwrite`` for ``.png``: (H, W, 3 | 4) uint8 / uint16 in (B, G, R[, A]) order."""
    from .flowou_codec import cv2_imencode_png
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    Path(path).write_bytes(cv2_imencode_png(np.asarray(img)).tobytes())


class GeneralVideoCapture(object):
    """A cv2.VideoCapture replacement that can also read images in a directory (and frame arrays)."""

    def __init__(self, path, reverse=False):
        path = Path(path)
        self.image_inputs = path.is_dir()
        self.array = None
        self.cap = None
        self.i = 0
        if self.image_inputs:
            self.path = path
            self.images = sorted([f for f in next(os.walk(path))[2] if os.path.splitext(f)[1].lower() in IMAGE_SUFFIXES])
            if reverse:
                self.images = self.images[::-1]
        elif path.suffix.lower() in ('.npy', '.npz'):
            arr = np.load(path)
            if not isinstance(arr, np.ndarray):
                arr = arr[arr.files[0]]
            if arr.ndim != 4 or arr.shape[-1] != 3 or arr.dtype != np.uint8:
                raise ValueError("frame array must be [T, H, W, 3] uint8 (BGR)")
            self.array = arr[::-1] if reverse else arr
        else:
            cv2 = _cv2()
            if cv2 is None:
                raise RuntimeError(f"cannot open {path}: video containers need OpenCV (cv2); give a directory of "
                                   "PNG frames or a .npy / .npz frame array instead")
            self.cap = cv2.VideoCapture(str(path))

    def read(self):
        if self.image_inputs:
            if self.i >= len(self.images):
                return False, None
            self.frame_src = self.images[self.i]
            img = imread_bgr(self.path / self.images[self.i])
            self.i += 1
            return True, img
        if self.array is not None:
            if self.i >= len(self.array):
                return False, None
            self.i += 1
            return True, np.ascontiguousarray(self.array[self.i - 1])
        return self.cap.read()

    def release(self):
        return None if self.cap is None else self.cap.release()


def get_video_frames(path):
    cap = GeneralVideoCapture(path)
    while True:
        success, frame = cap.read()
        if not success or frame is None:
            return None
        yield frame


def get_video_length(path):
    N = 0
    for _ in get_video_frames(path):
        N += 1
    return N


class FrameRing:
    """Host frames -> a rotation of PINNED host buffers, handed to the tracker as uint8 (H, W, 3) CPU tensors:

        for frame in FrameRing(get_video_frames(path)):
            meta = tracker.track(frame)

    The tracker's flow plugin uploads a pinned frame with a copy kernel (``mftx_copy_bytes``) on the stream that encodes it
    (the encoder side stream when ``C.async_encode`` is set): the host moves on to the next frame and the upload
    overlaps the previous frame's refinement.  (Round 2 blamed the SDMA queues for this ring and ``ResultDrain`` running at
    a third of the resident rate together; the cause was on the host: filling the pinned buffer with torch's ``copy_``
    ran a 786 kB memcpy on the whole intra-op thread pool, ``tools/io_paths3.py`` -- 45 vs 124 frames/s.)

    ``keep``: how many later frames a yielded frame stays valid for (its buffer is recycled after that); the default
    covers the tracker's memory ring (32 frames + the frames in flight).  ``streams``: the HIP streams the consumer
    uploads on (default: the stream current when the next frame is asked for; with ``C.async_encode`` pass the flow
    plugin's ``ensure_encode_stream()``).  A buffer is only overwritten once the uploads enqueued on those streams
    while it was the newest frame have completed -- an event recorded when the consumer comes back for the next
    frame; in steady state it completed dozens of frames ago and the check costs nothing.
    """

    def __init__(self, frames, keep=40, streams=None):
        self.frames = iter(frames)
        self.slots = max(2, int(keep))
        self.streams = streams
        self._pinned, self._events = [], {}

    def prepare(self, shape):
        """Pin all the buffers now (pinning is slow -- milliseconds each -- and otherwise happens frame by frame)."""
        while len(self._pinned) < self.slots:
            self._pinned.append(torch.empty(tuple(shape), dtype=torch.uint8).pin_memory())
        return self

    def __iter__(self):
        for n, frame in enumerate(self.frames):
            frame = np.ascontiguousarray(frame)
            if len(self._pinned) < self.slots:
                self._pinned.append(torch.empty(frame.shape, dtype=torch.uint8).pin_memory())
            buf = self._pinned[n % self.slots]
            if tuple(buf.shape) != frame.shape:
                raise ValueError("all frames of a video must have the same size")
            for ev in self._events.pop(n % self.slots, ()):          # uploads of the frame this buffer held before
                ev.synchronize()
            # a plain memcpy: torch's CPU copy_ would run on the whole intra-op thread pool (128 threads on the GPU box) -- 4-7 ms
            # for 786 kB, and its spinning workers then slow every host-side wait of the loop down (tools/io_paths3.py)
            np.copyto(buf.numpy(), frame)
            yield buf
            if torch.cuda.is_available():                            # the consumer is back: its uploads of `buf` are enqueued
                evs = []
                for st in (self.streams or [torch.cuda.current_stream()]):
                    ev = torch.cuda.Event()
                    ev.record(st)
                    evs.append(ev)
                self._events[n % self.slots] = evs


class ResultDrain:
    """Device results -> pinned host memory without stalling the loop: ``submit`` enqueues non-blocking copies of a
    result's three planes on the CALLER's stream -- right behind the kernels that produce them, 4 MB = ~0.1 ms at
    512 x 512 -- and returns immediately; ``collect`` waits for the oldest one and hands back its CPU tensors, so a
    loop that collects a couple of frames late never waits for the GPU.  (A separate download stream was measured
    and dropped: next to the pinned uploads it serialised the whole loop, 36 instead of 62 frames/s,
    ``tools/io_paths.py``.)  The pinned buffers form a ring of ``depth`` sets (pinning memory is slow, it happens
    once): a collected result stays valid until ``depth`` more results have been submitted (``copy=True`` returns
    private copies instead)."""

    def __init__(self, device="cuda", depth=4, check=None, nonfinite_from=None):
        """nonfinite_from: a tracker (or flow plugin) whose device-side non-finite counters ride along with every result: ``submit``
        enqueues a 16-byte copy of each counter into a pinned word behind the result's planes, ``collect`` reads the word after
        the event it waits for anyway and raises FloatingPointError -- no extra synchronisation, the pipeline stays asynchronous.
        check: optional callable run by every ``collect`` after its wait.  NOTE: ``tracker.check_nonfinite`` here reads the
        counter with ``.item()``, which waits for EVERYTHING queued on the stream (frames t+1 .. t+depth included), i.e. it
        drains the pipeline on every collect; prefer ``nonfinite_from``."""
        self.device = torch.device(device)
        self.depth = depth
        self.check = check
        src = getattr(nonfinite_from, "flower", nonfinite_from)
        self._nf_src = src if hasattr(src, "nonfinite_snapshot") else None
        self._nf_words = []            # per slot: pinned int32 [8, 4]
        self._sets, self._queue, self._n = [], [], 0

    def prepare(self, result):
        """Pin all `depth` buffer sets now, for results shaped like `result` (pinning is slow -- milliseconds per buffer --
        and otherwise happens inside the loop, during the first `depth` submits)."""
        planes = result.planes() if hasattr(result, "planes") else tuple(result)
        while len(self._sets) < self.depth:
            self._sets.append([torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in planes])
        return self

    def submit(self, result):
        planes = result.planes() if hasattr(result, "planes") else tuple(result)
        if len(self._queue) >= self.depth:
            raise RuntimeError("ResultDrain: collect() before submitting more than `depth` results")
        slot = self._n % self.depth
        if len(self._sets) <= slot:
            self._sets.append([torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in planes])
        host = self._sets[slot]
        from . import ops
        for h, t in zip(host, planes):
            # a copy KERNEL on the caller's stream, not hipMemcpyAsync: a pinned download in the SDMA queue holds back the
            # pinned uploads submitted behind it until the compute it waits for is done (profiles/r2_io_paths.txt)
            t = t.contiguous()
            if (t.is_cuda or t.is_pinned()) and t.data_ptr() % 16 == 0 and h.data_ptr() % 16 == 0:
                ops.copy_bytes(t, h)
            else:       # pageable host results (keep_result_on_device = False) or a plane view off the 16-byte grid: torch's copy
                h.copy_(t, non_blocking=True)
        words = None
        if self._nf_src is not None:
            rows = len(self._nf_src._all_engines()) if hasattr(self._nf_src, "_all_engines") else 8     # one row per engine: the
            while len(self._nf_words) <= slot:                       # plugin's split parts and lanes are user-configurable
                self._nf_words.append(None)
            if self._nf_words[slot] is None or self._nf_words[slot].shape[0] < rows:
                self._nf_words[slot] = torch.zeros(max(8, rows), 4, dtype=torch.int32).pin_memory()
            words = self._nf_words[slot]
            self._nf_src.nonfinite_snapshot(words)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._queue.append((ev, host, words))
        self._n += 1

    def collect(self, copy=False):
        ev, host, words = self._queue.pop(0)
        ev.synchronize()
        if words is not None:
            bad = int(words[:, 0].sum())
            if bad:       # (the counters stay set: every later collect raises too, until nonfinite_count(reset=True))
                raise self._nf_src.nonfinite_error(bad)
        if self.check is not None:
            self.check()
        return tuple(torch.from_numpy(h.numpy().copy()) for h in host) if copy else tuple(host)      # (numpy: a plain memcpy, see FrameRing)

    def __len__(self):
        return len(self._queue)
