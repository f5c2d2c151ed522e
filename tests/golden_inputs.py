"""Seeded INPUT builders shared by ``tools/make_goldens.py`` (which feeds them to
the reference in the build container) and by the tests (which feed the same
inputs to the oracle / the HIP path).  Only the reference's OUTPUTS are stored
under ``tests/golden/``; inputs are regenerated from these seeds.
"""
from __future__ import annotations

import numpy as np

OPS_H, OPS_W = 16, 24          # 1/8-resolution grid of the per-op fixtures (image 128x192)
SEQ_H, SEQ_W = 64, 96          # stub-flow tracking sequence
SEQ_FRAMES = 44
E2E_H, E2E_W = 128, 128        # real-RAFT tracking sequence
E2E_FRAMES = 42
E2E_ITERS = 4
WEIGHT_SEED = 7


def _rng(*key):
    return np.random.Generator(np.random.PCG64([0xC0FFEE, *key]))


def smooth_field(rng, C, H, W, cells=6, amp=1.0):
    """Low-frequency random field (C,H,W): bilinear upsample of a coarse grid."""
    g = rng.standard_normal((C, cells + 1, cells + 1)).astype(np.float32) * np.float32(amp)
    ty = np.linspace(0, cells, H, endpoint=False, dtype=np.float32)
    tx = np.linspace(0, cells, W, endpoint=False, dtype=np.float32)
    y0 = np.floor(ty).astype(np.int64); fy = (ty - y0)[None, :, None]
    x0 = np.floor(tx).astype(np.int64); fx = (tx - x0)[None, None, :]
    rows = g[:, y0] * (1 - fy) + g[:, y0 + 1] * fy
    return (rows[:, :, x0] * (1 - fx) + rows[:, :, x0 + 1] * fx).astype(np.float32)


def ops_inputs():
    """Inputs of the per-op fixtures at the 1/8 grid."""
    h, w = OPS_H, OPS_W
    r = _rng(1)
    d = {}
    d["fmap1"] = r.standard_normal((1, 256, h, w)).astype(np.float32)
    d["fmap2"] = r.standard_normal((1, 256, h, w)).astype(np.float32)
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    grid = np.stack([xs, ys])[None]
    flow = smooth_field(r, 2, h, w, cells=3, amp=4.0)[None]
    flow[0, :, 0, 0] = (-9.3, 2.25)      # far outside on the left
    flow[0, :, 3, 5] = (0.0, 0.0)        # exactly integral coordinates
    flow[0, :, 7, 11] = (30.5, -20.5)    # far outside
    flow[0, :, h - 1, w - 1] = (0.5, 0.5)
    d["coords1"] = (grid + flow).astype(np.float32)
    d["flow"] = flow.astype(np.float32)
    d["net"] = np.tanh(r.standard_normal((1, 128, h, w))).astype(np.float32)
    d["inp"] = np.maximum(r.standard_normal((1, 128, h, w)), 0).astype(np.float32)
    d["corr"] = r.standard_normal((1, 324, h, w)).astype(np.float32)
    d["delta_flow"] = (0.3 * r.standard_normal((1, 2, h, w))).astype(np.float32)
    d["motion"] = np.maximum(r.standard_normal((1, 128, h, w)), 0).astype(np.float32)
    d["mask"] = r.standard_normal((1, 576, h, w)).astype(np.float32)
    d["occl_lr"] = r.standard_normal((1, 2, h, w)).astype(np.float32)
    d["unc_lr"] = r.standard_normal((1, 1, h, w)).astype(np.float32)
    return d


def stub_flowou(left_id, right_id, H=SEQ_H, W=SEQ_W):
    """Deterministic fake FlowOU for the pair (left_id -> right_id): smooth flow
    proportional to the frame distance (so long chains leave the image), occlusion
    mostly below MFT's 0.02 threshold with blobs above it, positive sigma."""
    r = _rng(2, left_id + 1000, right_id + 1000)
    dt = float(right_id - left_id)
    flow = smooth_field(r, 2, H, W, cells=4, amp=0.9) * np.float32(dt) \
        + np.array([0.8 * dt, -0.45 * dt], np.float32)[:, None, None]
    o = smooth_field(r, 1, H, W, cells=5, amp=1.0)
    occl = np.where(o > 0.6, np.clip(o - 0.3, 0, 1), 0.015 * np.abs(np.tanh(o))).astype(np.float32)
    sigma = (0.05 + np.abs(smooth_field(r, 1, H, W, cells=5, amp=0.7))
             * np.float32(1 + 0.2 * np.log2(1 + abs(dt)))).astype(np.float32)
    if (left_id + right_id) % 5 == 0:          # a fully occluded corner: all candidates -inf
        occl[:, : H // 8, : W // 8] = 0.9
    return flow.astype(np.float32), occl, sigma


def id_image(frame_id, H=SEQ_H, W=SEQ_W):
    """uint8 BGR image that encodes its frame id (so a stub flower can recover
    which pair it is asked for)."""
    img = np.zeros((H, W, 3), np.uint8)
    img[0, 0, 0] = frame_id % 256
    img[0, 0, 1] = frame_id // 256
    return img


def decode_id(img):
    return int(img[0, 0, 0]) + 256 * int(img[0, 0, 1])


def checksum(t):
    a = np.asarray(t, dtype=np.float64)
    return np.array([a.sum(), np.abs(a).sum(), (a * a).sum()], np.float64)


def tapvid_inputs():
    """Random TAP-Vid style ground truth and predictions: 2 videos x 12 tracks x 20 frames on the
    256 x 256 raster (tracks = smooth random walks, ~25 % occluded, two never-visible tracks),
    predictions = ground truth + noise of mixed magnitude, ~10 % occlusion flips."""
    r = _rng(31)
    b, n, T = 2, 12, 20
    start = r.uniform(20, 236, size=(b, n, 1, 2))
    steps = r.normal(0, 2.0, size=(b, n, T, 2))
    gt_tracks = start + np.cumsum(steps, axis=2)
    gt_occluded = r.random((b, n, T)) < 0.25
    gt_occluded[:, 3] = True                     # never visible
    gt_occluded[0, 0, :4] = True                 # first visible late
    gt_occluded[1, 0, 0] = False
    noise = r.normal(0, 1.0, size=(b, n, T, 2)) * r.choice([0.3, 1.5, 5.0, 20.0], size=(b, n, T, 1))
    pred_tracks = gt_tracks + noise
    flip = r.random((b, n, T)) < 0.10
    pred_occluded = np.logical_xor(gt_occluded, flip)
    frames = r.integers(0, 255, size=(T, 8, 8, 3), dtype=np.uint8)
    return dict(gt_tracks=gt_tracks, gt_occluded=gt_occluded, pred_tracks=pred_tracks,
                pred_occluded=pred_occluded, frames=frames)


def tapvid_pickle_sequences():
    """TAP-Vid-shaped sequences for the dataset-reader pin: {name: {'video' (N, H, W, 3) uint8 RGB smooth texture,
    'points' (n, N, 2) xy in [0, 1], 'occluded' (n, N) bool}} -- two sequences of different size / aspect ratio (the real
    pickles hold 256-frame DAVIS clips; these are 8 frames of 40 x 56 and 36 x 36)."""
    r = _rng(53)
    out = {}
    for name, (N, H, W, n) in (("seq-wide", (8, 40, 56, 7)), ("seq-square", (8, 36, 36, 5))):
        base = smooth_field(r, 3, H + 16, W + 16, cells=4, amp=110.0) + 128.0
        video = np.stack([np.clip(base[:, i: i + H, i // 2: i // 2 + W], 0, 255).transpose(1, 2, 0) for i in range(N)]).astype(np.uint8)
        pts = np.clip(r.uniform(0.05, 0.95, size=(n, 1, 2)) + np.cumsum(r.normal(0, 0.01, size=(n, N, 2)), axis=1), 0.0, 0.999)
        occ = r.random((n, N)) < 0.25
        occ[0] = True                               # a never-visible track
        occ[1, :3] = True                           # first visible late
        occ[2, 0] = False
        out[name] = {"video": np.ascontiguousarray(video), "points": pts.astype(np.float64), "occluded": occ}
    return out


TAPVID_SCALINGS = ("default", "false", "fullres", "32x24_64x48", "x30", "48x", "fullres_40x40", "256x256_512x512")


def results_api_inputs():
    """A FlowOU result on a 40 x 56 frame (smooth flow that leaves the frame on two sides), an image to warp,
    a mask, and query points incl. out-of-frame and integral ones."""
    H, W = 40, 56
    r = _rng(41)
    flow = smooth_field(r, 2, H, W, cells=3, amp=6.0)
    flow[:, :, :6] -= 9.0                       # far left columns leave the frame
    occl = (r.random((1, H, W)) * 0.3).astype(np.float32)
    sigma = (0.2 + r.random((1, H, W))).astype(np.float32)
    img = r.random((H, W, 3)).astype(np.float32)
    mask = r.random((H, W)) < 0.7
    pts = np.array([[0.0, 0.0], [3.25, 7.5], [55.0, 39.0], [-2.0, 4.0], [60.5, 10.0], [27.0, 20.0], [10.75, 38.6]],
                   np.float32)
    return dict(flow=flow.astype(np.float32), occl=occl, sigma=sigma, img=img, mask=mask, pts=pts)


def codec_inputs():
    """One FlowOU triple for the flow-cache codec pin (MFT/utils/io.py:495-563): 37 x 53 (H*W odd, so plane
    views are not 16-byte aligned), a constant channel (the ub == lb branch) and values that hit both ends
    of the uint16 range."""
    H, W = 37, 53
    r = _rng(51)
    flow = smooth_field(r, 2, H, W, cells=4, amp=7.0)
    occl = np.clip(smooth_field(r, 1, H, W, cells=5, amp=0.6), 0, 1).astype(np.float32)
    sigma = np.full((1, H, W), 0.75, np.float32)           # constant plane -> all zeros after quantisation
    return dict(flow=flow.astype(np.float32), occl=occl, sigma=sigma)


def init_flow_input(H, W):
    """Full-resolution initial flow for compute_flow(init_flow=...) (MFT/raft.py:49-52)."""
    return smooth_field(_rng(61), 2, H, W, cells=3, amp=5.0).astype(np.float32)
