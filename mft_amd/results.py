"""FlowOUTrackingResult -- drop-in for ``MFT/results.py:11-265``.

Same constructor (with the reference's assertions), same attributes
(``flow [2,H,W]``, ``occlusion [1,H,W]``, ``sigma [1,H,W]``, ``H``, ``W``) and
the same public methods.  On device tensors ``chain`` / ``warp_backward`` run
on libmftx's HIP kernels (one fused launch instead of three ``grid_sample``
pipelines); the host-side, few-hundred-point helpers (``sample``,
``warp_forward_points``) stay torch plumbing.
"""
from __future__ import annotations

import pickle

import numpy as np
import torch
import torch.nn.functional as F

from . import ops


def _grid(H, W, device):
    """x,y pixel grid [2,H,W] (MFT/utils/geom_utils.py:429-452)."""
    idx = torch.arange(H * W, device=device)
    return torch.stack([idx % W, torch.div(idx, W, rounding_mode="floor")], 0).reshape(2, H, W).to(torch.float32)


def _normalize_coords(coords, H, W):
    """(N H W xy) pixel coords -> [-1,1] (MFT/utils/interpolation.py:63-73)."""
    scales = torch.from_numpy(np.array([2 / (W - 1), 2 / (H - 1)]).astype(np.float32)).to(coords.device)
    return coords * scales.reshape(1, 1, 1, 2) - 1


class FlowOUTrackingResult(object):
    def __init__(self, flow, occlusion=None, sigma=None, validate=True):
        """Stores optical flow, occlusion map and flow sigma map.

        flow: (xy-delta, H, W) tensor; occlusion, sigma: (1, H, W) tensors.
        ``validate=False`` skips the three range assertions (each is a host
        sync); the tracker uses it for tensors its own kernels just produced.
        """
        assert len(flow.shape) == 3
        assert flow.shape[0] == 2
        self.H, self.W = flow.shape[1:]
        if occlusion is None:
            occlusion = torch.zeros((1, self.H, self.W), dtype=torch.float32)
        if sigma is None:
            sigma = torch.zeros((1, self.H, self.W), dtype=torch.float32)

        assert flow.shape == (2, self.H, self.W)
        assert occlusion.shape == (1, self.H, self.W)
        assert sigma.shape == (1, self.H, self.W)

        if validate:
            assert torch.all(occlusion >= 0)
            assert torch.all(occlusion <= 1.000001)
            assert torch.all(sigma >= 0)

        self.flow = flow
        self.occlusion = occlusion
        self.sigma = sigma

    def __repr__(self):
        return f'<{self.__class__.__name__} ({self.H} x {self.W}) has flow, occlusion, sigma>'

    # ---- placement -------------------------------------------------------
    def cpu(self):
        self.flow, self.occlusion, self.sigma = self.flow.cpu(), self.occlusion.cpu(), self.sigma.cpu()
        return self

    def cuda(self):
        self.flow, self.occlusion, self.sigma = self.flow.cuda(), self.occlusion.cuda(), self.sigma.cuda()
        return self

    def clone(self):
        return FlowOUTrackingResult(self.flow.clone(), self.occlusion.clone(), self.sigma.clone(), validate=False)

    def planes(self):
        return self.flow, self.occlusion, self.sigma

    # ---- IO (MFT/results.py:61-72): the reference's .flowouX16 cache entry, quantised on the device
    # (mft_amd/flowou_codec.py); any other name is a plain fp32 pickle of the three arrays ------
    def write(self, path):
        """``*.flowou.png`` / ``*.flowouX16.pkl`` / ``*.flowouX32.pkl``: the reference's cache-entry formats
        (MFT/results.py:61-65 -> MFT/utils/io.py:174-197; X16 is quantised on the device); any other name: a plain
        fp32 pickle."""
        from pathlib import Path
        suffixes = Path(path).suffixes
        if suffixes and suffixes[0] in (".flowou", ".flowouX16", ".flowouX32"):
            from .flowou_codec import write_flowou
            write_flowou(path, self.flow, self.occlusion, self.sigma)
            return
        with open(path, "wb") as f:
            pickle.dump({k: getattr(self, k).detach().cpu().numpy() for k in ("flow", "occlusion", "sigma")}, f)

    @classmethod
    def read(cls, path):
        """Counterpart of ``write``; like the reference (MFT/results.py:67-72) the result is on the host."""
        from pathlib import Path
        suffixes = Path(path).suffixes
        if suffixes and suffixes[0] in (".flowou", ".flowouX16", ".flowouX32"):
            from .flowou_codec import read_flowou
            return FlowOUTrackingResult(*(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)) for a in read_flowou(path)))
        with open(path, "rb") as f:
            d = pickle.load(f)
        return FlowOUTrackingResult(torch.from_numpy(d["flow"]), torch.from_numpy(d["occlusion"]),
                                    torch.from_numpy(d["sigma"]))

    @classmethod
    def identity(cls, flow_shape, device=None):
        """Zero-flow, zero-sigma, zero-occlusion result; flow_shape = (H, W)."""
        H, W = flow_shape
        return FlowOUTrackingResult(torch.zeros((2, H, W), dtype=torch.float32, device=device),
                                    torch.zeros((1, H, W), dtype=torch.float32, device=device),
                                    torch.zeros((1, H, W), dtype=torch.float32, device=device), validate=False)

    # ---- chaining (hot path) ---------------------------------------------
    def chain(self, flow):
        """With self.flow A->B and ``flow`` B->C, the flow A->C by bilinear
        interpolation (MFT/results.py:87-114)."""
        assert flow.shape == (2, self.H, self.W)
        flow = flow.to(torch.float32).contiguous()
        zeros = torch.zeros((1, self.H, self.W), dtype=torch.float32, device=flow.device)
        out = ops.chain((self.flow.to(flow.device).to(torch.float32).contiguous(), zeros, zeros),
                        (flow, zeros, zeros))
        return out[0]

    def warp_backward(self, img):
        """Sample img (C,H,W) at the right end of self.flow (MFT/results.py:116-136)."""
        assert len(img.shape) == 3
        assert img.shape[1:] == (self.H, self.W)
        return ops.warp_backward(self.flow.to(img.device).to(torch.float32).contiguous(),
                                 img.to(torch.float32).contiguous())

    def chain_result(self, right):
        """chain_results(self, right) (MFT/MFT.py:233-239) in one kernel."""
        flow, occl, sigma = ops.chain(self.planes(), right.planes())
        return FlowOUTrackingResult(flow, occl, sigma, validate=False)

    # ---- point queries (N ~ hundreds; torch plumbing) ----------------------
    def _points_normed(self, points):
        if not isinstance(points, torch.Tensor):
            points = torch.from_numpy(np.asarray(points))
        points = points.to(torch.float32)
        return points, _normalize_coords(points.reshape(1, 1, -1, 2), self.H, self.W)

    def warp_forward_points(self, points):
        """(N xy) source coords -> (N xy) warped coords (MFT/results.py:138-157)."""
        points, normed = self._points_normed(points)
        flow = self.flow.to(points.device).to(torch.float32)
        s = F.grid_sample(flow[None], normed, align_corners=True)
        return points + s[0, :, 0, :].t()

    def sample(self, points):
        """-> sampled flow (xy N), occlusion (1 N), sigma (1 N) (MFT/results.py:159-188)."""
        points, normed = self._points_normed(points)
        dev = points.device
        out = []
        for t in (self.flow, self.occlusion, self.sigma):
            out.append(F.grid_sample(t.to(dev).to(torch.float32)[None], normed, align_corners=True)[0, :, 0, :])
        return tuple(out)

    def warp_forward(self, img, mask=None, border=None):
        """Forward-splat ``img`` (H, W, ...) along the flow: every (unmasked) source pixel spreads its value
        over the four pixels around its destination with bilinear weights (destination and corners
        clamped into the image), and each output pixel is the weight-normalised sum of what reached it --
        MFT/results.py:190-248 with MFT/utils/interpolation.py:234-309.  Pixels nothing reached are 0, or
        ``border``.  Returns a numpy array like the reference.  (Visualisation helper: torch ops on
        whatever device the flow lives on; not part of the tracking path.)"""
        dev = self.flow.device
        H, W = self.H, self.W
        assert tuple(img.shape[:2]) == (H, W)
        vals = torch.as_tensor(np.asarray(img) if not isinstance(img, torch.Tensor) else img).to(dev)
        extra = vals.shape[2:]
        vals = vals.reshape(H * W, -1)
        vmin, vmax = vals.min(), vals.max()
        dst = (_grid(H, W, dev) + self.flow.to(torch.float32)).reshape(2, H * W)
        if mask is not None:
            keep = torch.as_tensor(np.asarray(mask) if not isinstance(mask, torch.Tensor) else mask).to(dev).reshape(-1).bool()
            dst, vals = dst[:, keep], vals[keep]
        x, y = dst[0], dst[1]
        x0, y0 = torch.floor(x).long(), torch.floor(y).long()
        x1, y1 = x0 + 1, y0 + 1
        x, y = x.clamp(0, W - 1), y.clamp(0, H - 1)
        x0, x1 = x0.clamp(0, W - 1), x1.clamp(0, W - 1)
        y0, y1 = y0.clamp(0, H - 1), y1.clamp(0, H - 1)
        wx0, wx1 = x1.float() - x, x - x0.float()
        wy0, wy1 = y1.float() - y, y - y0.float()
        idx = torch.cat((y0 * W + x0, y1 * W + x0, y0 * W + x1, y1 * W + x1))
        wts = torch.cat((wx0 * wy0, wx0 * wy1, wx1 * wy0, wx1 * wy1))[:, None]
        vals = vals.to(wts.dtype) if not vals.is_floating_point() else vals
        accum = torch.zeros((H * W, vals.shape[1]), dtype=vals.dtype, device=dev)
        accum = accum.index_put((idx,), vals.repeat(4, 1) * wts.to(vals.dtype), accumulate=True)
        counts = torch.zeros((H * W, 1), dtype=wts.dtype, device=dev).index_put((idx,), wts, accumulate=True)
        hit = counts[:, 0] > 0
        out = accum.clone()
        out[hit] /= counts[hit].to(out.dtype)
        eps = 5e-3
        assert out.min() >= min(float(vmin), 0) - eps and out.max() <= max(float(vmax), 0) + eps
        if border is not None:
            out[~hit] = border
        return out.reshape(H, W, *extra).cpu().numpy()

    def invalid_mask(self):
        """(H, W) bool, True where the flow points outside the image
        (MFT/results.py:250-265)."""
        q = _grid(self.H, self.W, self.flow.device) + self.flow.to(torch.float32)
        return (q[0] < 0) | (q[1] < 0) | (q[0] >= self.W) | (q[1] >= self.H)


class PendingHostResult(FlowOUTrackingResult):
    """A ``FlowOUTrackingResult`` on the HOST whose planes are still on their way: what ``MFT.track()`` returns as ``meta.result``
    (MFT/MFT.py:145-148 returns a CPU result from every call) without making every call wait for the GPU.

    The three planes are views of one pinned host buffer ``[4, H, W]`` that a copy kernel, enqueued behind the frame's selection
    kernel, fills; the first access to ``flow`` / ``occlusion`` / ``sigma`` (or any method, ``clone``, ``cpu``, pickling) waits
    for the event behind that copy -- once -- and from then on this is an ordinary CPU result.  A caller that reads every result
    right away (the reference's demo.py:59-65) synchronises per frame exactly as with the reference; a caller that collects
    results and reads them later (a runner that writes its outputs at the end, a consumer thread) lets the tracker run ahead and
    gets the pipelined rate.  ``ready()`` asks without waiting.

    The non-finite guard of the flow plugin rides along: a 16-byte snapshot of every engine's counter lands in pinned words
    behind the planes, and the first access raises ``FloatingPointError`` if any is set (``on_wait``)."""

    def __init__(self, host, event, on_wait=None):
        assert host.dim() == 3 and host.shape[0] == 4 and not host.is_cuda
        self.H, self.W = host.shape[1:]
        self._host, self._event, self._on_wait = host, event, on_wait
        self._flow, self._occlusion, self._sigma = host[0:2], host[2:3], host[3:4]

    def ready(self):
        """True once the planes have arrived (never waits)."""
        return self._event is None or self._event.query()

    def wait(self):
        ev = self._event
        if ev is not None:
            ev.synchronize()
            self._event = None
            cb, self._on_wait = self._on_wait, None
            if cb is not None:
                cb()
        return self

    def _get(name):
        def getter(self):
            self.wait()
            return getattr(self, name)

        def setter(self, value):
            setattr(self, name, value)
        return property(getter, setter)

    flow = _get("_flow")
    occlusion = _get("_occlusion")
    sigma = _get("_sigma")
    del _get

    def __repr__(self):
        return f'<{self.__class__.__name__} ({self.H} x {self.W}) has flow, occlusion, sigma{"" if self.ready() else " (in flight)"}>'

    def __reduce__(self):          # pickles (and deep-copies) as the plain CPU result it stands for
        self.wait()
        return (_rebuild_result, (self._flow.clone(), self._occlusion.clone(), self._sigma.clone()))


def _rebuild_result(flow, occlusion, sigma):
    return FlowOUTrackingResult(flow, occlusion, sigma, validate=False)


FlowOUResult = FlowOUTrackingResult  # the name BASELINE.json uses
