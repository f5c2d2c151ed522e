"""Seeded synthetic video for parity tests and benchmarks.

There is no dataset, no cv2 and no network in the build environment, so the
``demo_in`` video of the reference (``demo.py:59``) is replaced by a
deterministic stand-in of the same kind of content: a smooth multi-octave random
texture that translates, rotates and zooms slowly, with one independently
moving occluder.  Frames are uint8 ``H x W x 3`` in BGR order, i.e. exactly what
``MFT.init()/MFT.track()`` take (``MFT/MFT.py:22-66``).
"""
from __future__ import annotations

import numpy as np


def _smooth_noise(rng, size, octaves=(8, 24, 64)):
    """Sum of bilinearly upsampled white-noise grids -> (size, size, 3) in [0,1]."""
    acc = np.zeros((size, size, 3), np.float32)
    amp_total = 0.0
    for cells in octaves:
        g = rng.random((cells + 1, cells + 1, 3), dtype=np.float32)
        t = np.linspace(0, cells, size, endpoint=False, dtype=np.float32)
        i0 = np.floor(t).astype(np.int64)
        f = (t - i0).astype(np.float32)
        rows = g[i0] * (1 - f)[:, None, None] + g[i0 + 1] * f[:, None, None]
        img = rows[:, i0] * (1 - f)[None, :, None] + rows[:, i0 + 1] * f[None, :, None]
        amp = 1.0 / np.sqrt(cells)
        acc += amp * img
        amp_total += amp
    acc /= amp_total
    acc -= acc.min()
    acc /= max(float(acc.max()), 1e-6)
    return acc


def _sample_bilinear(canvas, xs, ys):
    """Bilinear sample (clamped) of canvas[S,S,3] at float coords."""
    S = canvas.shape[0]
    xs = np.clip(xs, 0, S - 1.001)
    ys = np.clip(ys, 0, S - 1.001)
    x0 = np.floor(xs).astype(np.int64)
    y0 = np.floor(ys).astype(np.int64)
    fx = (xs - x0)[..., None].astype(np.float32)
    fy = (ys - y0)[..., None].astype(np.float32)
    a = canvas[y0, x0]
    b = canvas[y0, x0 + 1]
    c = canvas[y0 + 1, x0]
    d = canvas[y0 + 1, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


class SyntheticVideo:
    """Deterministic frame source: ``video[i]`` -> uint8 BGR ``(H, W, 3)``."""

    def __init__(self, height=512, width=512, n_frames=200, seed=0):
        self.H, self.W, self.n_frames, self.seed = height, width, n_frames, seed
        rng = np.random.Generator(np.random.PCG64([seed, 0xF00D]))
        self._S = int(1.25 * max(height, width)) + 2
        self._canvas = _smooth_noise(rng, self._S)
        self._occ = _smooth_noise(rng, 64, octaves=(4, 16))
        ph = rng.uniform(0, 2 * np.pi, size=6)
        self._ph = ph
        yy, xx = np.meshgrid(np.arange(height, dtype=np.float32),
                             np.arange(width, dtype=np.float32), indexing="ij")
        self._xx, self._yy = xx - width / 2, yy - height / 2

    def __len__(self):
        return self.n_frames

    def camera(self, i):
        """Frame i's camera: canvas = A @ (pixel - centre) + b  ->  (A (2,2), b (2,))."""
        H, W, S, ph = self.H, self.W, self._S, self._ph
        t = float(i)
        # sub-pixel drift, slow rotation and zoom
        tx = 0.08 * max(H, W) * np.sin(0.031 * t + ph[0]) + 0.37 * t * 0.1
        ty = 0.06 * max(H, W) * np.sin(0.023 * t + ph[1])
        ang = 0.10 * np.sin(0.017 * t + ph[2])
        zoom = 1.0 + 0.08 * np.sin(0.013 * t + ph[3])
        ca, sa = np.cos(ang) * zoom, np.sin(ang) * zoom
        return np.array([[ca, -sa], [sa, ca]]), np.array([S / 2 + tx, S / 2 + ty])

    def occluder(self, i):
        """Frame i's occluding square: (x0, y0, side) in pixels (may stick out of the frame)."""
        H, W, ph = self.H, self.W, self._ph
        t = float(i)
        side = max(H, W) // 6
        cx = W / 2 + 0.3 * W * np.sin(0.05 * t + ph[4])
        cy = H / 2 + 0.3 * H * np.cos(0.04 * t + ph[5])
        return int(round(cx - side / 2)), int(round(cy - side / 2)), side

    def __getitem__(self, i):
        if not 0 <= i < self.n_frames:
            raise IndexError(i)
        H, W = self.H, self.W
        (ca, msa), (sa, _) = self.camera(i)[0]
        bx, by = self.camera(i)[1]
        xs = ca * self._xx + msa * self._yy + bx
        ys = sa * self._xx + ca * self._yy + by
        img = _sample_bilinear(self._canvas, xs, ys)
        # occluder: textured square on its own trajectory
        x0, y0, side = self.occluder(i)
        xa, xb = max(x0, 0), min(x0 + side, W)
        ya, yb = max(y0, 0), min(y0 + side, H)
        if xb > xa and yb > ya:
            u = ((np.arange(xa, xb) - x0) * 63.0 / side).astype(np.float32)
            v = ((np.arange(ya, yb) - y0) * 63.0 / side).astype(np.float32)
            uu, vv = np.meshgrid(u, v)
            patch = _sample_bilinear(self._occ, uu, vv)
            img[ya:yb, xa:xb] = 0.25 + 0.75 * patch[..., ::-1]
        return np.ascontiguousarray((img * 255.0 + 0.5).clip(0, 255).astype(np.uint8))

    def ground_truth_tracks(self, points_xy, frame_q):
        """Where the BACKGROUND points seen at ``points_xy`` (n, 2) in frame ``frame_q`` are in every
        frame: ``tracks`` (n, n_frames, 2) xy pixels and ``occluded`` (n, n_frames) bool -- true
        when the point is behind the occluding square or outside the frame.  (A point that sits on
        the square in the query frame is tracked as the background behind it.)"""
        pts = np.asarray(points_xy, np.float64).reshape(-1, 2)
        ctr = np.array([self.W / 2, self.H / 2])
        A, b = self.camera(frame_q)
        canvas = (pts - ctr) @ A.T + b
        tracks = np.zeros((len(pts), self.n_frames, 2))
        occluded = np.zeros((len(pts), self.n_frames), bool)
        for t in range(self.n_frames):
            A, b = self.camera(t)
            p = (canvas - b) @ np.linalg.inv(A).T + ctr
            x0, y0, side = self.occluder(t)
            behind = (p[:, 0] >= x0 - 0.5) & (p[:, 0] < x0 + side - 0.5) & (p[:, 1] >= y0 - 0.5) & (p[:, 1] < y0 + side - 0.5)
            outside = (p[:, 0] < 0) | (p[:, 0] > self.W - 1) | (p[:, 1] < 0) | (p[:, 1] > self.H - 1)
            tracks[:, t] = p
            occluded[:, t] = behind | outside
        return tracks, occluded

    def frames(self, start=0, stop=None):
        for i in range(start, self.n_frames if stop is None else stop):
            yield self[i]
