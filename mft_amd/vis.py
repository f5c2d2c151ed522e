"""Overlays of the demo (``demo.py:116-146``) without OpenCV: tracked-point dots and the first-frame edit carried
along the tracked flow.  Visualisation only -- numpy on the host, the splatting itself is
``FlowOUTrackingResult.warp_forward`` (``MFT/results.py:190-248``)."""
from __future__ import annotations

import numpy as np
import torch

RED = (0, 0, 255)          # BGR, MFT/utils/vis_utils.py:27


def to_gray_3ch(img):
    """BGR uint8 -> 3-channel gray (cv2.cvtColor BGR2GRAY -> GRAY2BGR, MFT/utils/vis_utils.py:240-242):
    Y = 0.299 R + 0.587 G + 0.114 B in cv2's 14-bit fixed point (4899, 9617, 1868), rounded."""
    img = np.asarray(img).astype(np.int64)
    y = (img[..., 0] * 1868 + img[..., 1] * 9617 + img[..., 2] * 4899 + (1 << 13)) >> 14
    return np.repeat(y.astype(np.uint8)[..., None], 3, axis=2)


def blend_with_alpha_premult(img1_premult, img2, img1_alpha):
    """MFT/utils/vis_utils.py:755-765."""
    img1_alpha = np.asarray(img1_alpha)
    if img1_alpha.max() > 1.0001:
        img1_alpha = img1_alpha.astype(np.float32) / 255.0
    if img1_alpha.ndim == 2:
        img1_alpha = img1_alpha[..., None]
    result = np.asarray(img1_premult).astype(np.float32) + np.asarray(img2).astype(np.float32) * (1 - img1_alpha)
    return result.clip(0, 255).astype(np.uint8)


def draw_dots(frame, coords, occlusions, radius=3, color=RED):
    """A filled dot at every visible tracked point (demo.py:116-126; occlusion > 0.5 = hidden).
    coords (N, 2) xy, occlusions (N,)."""
    canvas = np.array(frame, copy=True)
    H, W = canvas.shape[:2]
    coords = coords.detach().cpu().numpy() if isinstance(coords, torch.Tensor) else np.asarray(coords)
    occl = occlusions.detach().cpu().numpy() if isinstance(occlusions, torch.Tensor) else np.asarray(occlusions)
    r = int(np.ceil(radius))
    dy, dx = np.mgrid[-r:r + 1, -r:r + 1]
    for (x, y), o in zip(coords.reshape(-1, 2), occl.reshape(-1)):
        if o > 0.5 or not (np.isfinite(x) and np.isfinite(y)):
            continue
        cx, cy = int(round(float(x))), int(round(float(y)))
        disc = (dx + cx - x) ** 2 + (dy + cy - y) ** 2 <= (radius + 0.5) ** 2
        ys, xs = dy[disc] + cy, dx[disc] + cx
        ok = (ys >= 0) & (ys < H) & (xs >= 0) & (xs < W)
        canvas[ys[ok], xs[ok]] = color
    return canvas


def draw_edit(frame, result, edit):
    """The RGBA first-frame edit (``cv2.imread(..., IMREAD_UNCHANGED)``: B, G, R, A) splatted along the tracked flow
    onto the gray current frame, for template pixels that are visible and inside the edit (demo.py:128-146)."""
    edit = np.asarray(edit)
    visible = (result.occlusion[0] < 0.5).cpu()
    mask = torch.logical_and(visible, torch.from_numpy(edit[:, :, 3] > 0))
    alpha = edit[:, :, 3:4].astype(np.float32) / 255.0
    premult = edit[:, :, :3].astype(np.float32) * alpha
    color = np.clip(np.asarray(result.warp_forward(premult, mask=mask)), 0, 255).astype(np.uint8)
    alpha_t = np.asarray(result.warp_forward(edit[:, :, 3:4], mask=mask))
    return blend_with_alpha_premult(color, to_gray_3ch(frame), alpha_t)


def get_queries(frame_shape, spacing):
    """Regular grid of query points, (N, 2) xy float32 (demo.py:105-114)."""
    H, W = frame_shape
    xs, ys = np.meshgrid(np.arange(0, W, spacing), np.arange(0, H, spacing))
    return torch.from_numpy(np.vstack((xs.flatten(), ys.flatten())).T).float()
