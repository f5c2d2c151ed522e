"""TAP-Vid protocol around the tracker: query sampling,
per-sequence runner and the benchmark metrics.

Reference behaviour restated here (nothing is imported from it):
  * ``MFT/runners/run_MFT_tapvid.py:139-237`` -- for every distinct query frame the
    tracker is re-initialised on that frame and run forward (and, in 'strided' mode,
    also backward) over the video, all runs of one sequence sharing ONE flow cache, so
    that every (left, right) flow pair is computed once; the dense result of every
    frame is read out at the query points (``MFT/point_tracking.py:6-27``); predicted
    tracks are rescaled to the 256 x 256 raster the metrics are defined on.
  * ``MFT/runners/run_MFT_tapvid.py:249-283`` -- ``track_sequence``.
  * ``MFT/evaluation/tapvid_eval_stuff.py:82-237`` -- occlusion accuracy,
    points-within-threshold and Jaccard at 1/2/4/8/16 px (TAP-Vid paper), plus the
    confusion counts and precision the reference adds.
  * ``MFT/evaluation/tapvid_eval_stuff.py:275-386`` -- 'strided' / 'first' query sampling.

The dataset readers (pickled DAVIS / Kinetics / RGB-stacking shards) are not
rebuilt: no dataset is reachable from the build environment.  ``synthetic_sequence``
makes TAP-Vid-shaped ground truth from the seeded synthetic video instead.
"""
from __future__ import annotations

import numpy as np
import torch

from .point_tracking import convert_to_point_tracking

THRESHOLDS = (1, 2, 4, 8, 16)


# ---------------------------------------------------------------------------
# query sampling
# ---------------------------------------------------------------------------
def sample_queries_strided(target_occluded, target_points, frames, query_stride=5):
    """Every ``query_stride``-th frame, every track visible there becomes a query.
    target_occluded (n_tracks, n_frames) bool, target_points (n_tracks, n_frames, 2) xy.
    Returns the batch-of-one dict of the reference: video, query_points (1, q, 3) as
    (t, y, x), target_points (1, q, n_frames, 2), occluded (1, q, n_frames), trackgroup (1, q)."""
    target_occluded = np.asarray(target_occluded)
    target_points = np.asarray(target_points)
    n_tracks, n_frames = target_occluded.shape
    t_idx, track_idx = [], []
    for t in range(0, n_frames, query_stride):
        vis = np.flatnonzero(target_occluded[:, t] == 0)
        t_idx.append(np.full(len(vis), t))
        track_idx.append(vis)
    t_idx, track_idx = np.concatenate(t_idx), np.concatenate(track_idx)
    xy = target_points[track_idx, t_idx]
    queries = np.stack([t_idx.astype(xy.dtype), xy[:, 1], xy[:, 0]], axis=-1)
    return {
        "video": np.asarray(frames)[None],
        "query_points": queries[None],
        "target_points": target_points[track_idx][None],
        "occluded": target_occluded[track_idx][None],
        "trackgroup": track_idx[None],
    }


def sample_queries_first(target_occluded, target_points, frames):
    """The first visible point of every track (never-visible tracks are dropped) is its query."""
    target_occluded = np.asarray(target_occluded)
    target_points = np.asarray(target_points)
    keep = (~target_occluded.astype(bool)).any(axis=1)
    occ, pts = target_occluded[keep], target_points[keep]
    first = np.argmax(occ == 0, axis=1)
    rows = np.arange(len(first))
    queries = np.stack([first.astype(pts.dtype), pts[rows, first, 1], pts[rows, first, 0]], axis=-1)
    return {
        "video": np.asarray(frames)[None],
        "query_points": queries.reshape(-1, 3)[None],
        "target_points": pts[None],
        "occluded": occ[None],
        "trackgroup": np.arange(len(first))[None],
    }


# ---------------------------------------------------------------------------
# metrics
# ---------------------------------------------------------------------------
def compute_tapvid_metrics(query_points, gt_occluded, gt_tracks, pred_occluded, pred_tracks, query_mode):
    """TAP-Vid metrics per video.  query_points (b, n, 3) as (t, y, x); gt_occluded /
    pred_occluded (b, n, T); gt_tracks / pred_tracks (b, n, T, 2) xy, all on the 256 x 256
    raster.  Returns a dict of (b,) arrays with the reference's keys."""
    if query_mode not in ("first", "strided"):
        raise ValueError("Unknown query mode " + query_mode)
    gt_occluded = np.asarray(gt_occluded)
    pred_occluded = np.asarray(pred_occluded)
    b, n, T = gt_occluded.shape
    qf = np.round(np.asarray(query_points)[..., 0]).astype(np.int32)
    ev = np.arange(T)[None, None, :] != qf[..., None]          # the query frame itself is not scored
    if query_mode == "first":
        # As in the reference, the loop runs over the BATCH axis and takes the first index at which
        # ANY entry of that video's occlusion table is visible, i.e. the first visible query ROW,
        # and masks the rows before it (tapvid_eval_stuff.py:147-151).
        for i in range(b):
            ev[i, : np.where(gt_occluded[i] == 0)[0][0]] = False

    def count(mask):
        return np.sum(mask & ev, axis=(1, 2))

    m = {"occlusion_accuracy": count(np.equal(pred_occluded, gt_occluded)) / np.sum(ev)}
    p_occ, p_vis = pred_occluded > 0.5, pred_occluded < 0.5
    g_occ, g_vis = gt_occluded > 0.5, gt_occluded < 0.5
    m["occlusion_FP"] = count(p_occ & g_vis)
    m["occlusion_FN"] = count(p_vis & g_occ)
    m["occlusion_TP"] = count(p_occ & g_occ)
    m["occlusion_TN"] = count(p_vis & g_vis)

    visible = np.logical_not(gt_occluded)
    pred_visible = np.logical_not(pred_occluded)
    d2 = np.sum(np.square(np.asarray(pred_tracks) - np.asarray(gt_tracks)), axis=-1)
    n_visible = count(visible)
    within_all, jac_all, prec_all = [], [], []
    for thr in THRESHOLDS:
        near = d2 < np.square(thr)
        hit = near & visible
        m[f"pts_within_{thr}"] = count(hit) / n_visible
        tp = count(hit & pred_visible)
        m[f"prec_at_{thr}"] = tp / count(pred_visible & visible)
        # Jaccard = TP / (TP + FN + FP); TP + FN is every visible ground-truth point, FP a point
        # predicted visible that is occluded or too far away
        fp = count(((~visible) & pred_visible) | ((~near) & pred_visible))
        m[f"jaccard_{thr}"] = tp / (n_visible + fp)
        within_all.append(m[f"pts_within_{thr}"])
        prec_all.append(m[f"prec_at_{thr}"])
        jac_all.append(m[f"jaccard_{thr}"])
    m["average_jaccard"] = np.mean(np.stack(jac_all, axis=1), axis=1)
    m["average_pts_within_thresh"] = np.mean(np.stack(within_all, axis=1), axis=1)
    m["average_prec"] = np.mean(np.stack(prec_all, axis=1), axis=1)
    return m


# ---------------------------------------------------------------------------
# runner
# ---------------------------------------------------------------------------
def track_sequence(tracker, video, start_frame, direction="forward", debug=False, flow_cache=None):
    """init on ``video[start_frame]``, then track to the end (or back to frame 0).
    Returns {frame_i: meta} with ``meta.frame_i`` / ``meta.backward`` set like the reference."""
    assert direction in ("forward", "backward")
    n_frames = len(video)
    if direction == "forward":
        frame_ids, time_direction = range(start_frame, n_frames), +1
    else:
        frame_ids, time_direction = range(start_frame, -1, -1), -1
    metas = {}
    for k, frame_i in enumerate(frame_ids):
        frame = video[frame_i]
        if k == 0:
            meta = tracker.init(frame, start_frame_i=start_frame, time_direction=time_direction,
                                flow_cache=flow_cache)
        else:
            try:
                meta = tracker.track(frame, debug=debug)
            except StopIteration:
                break
        meta.frame_i = frame_i
        meta.backward = direction == "backward"
        metas[frame_i] = meta
    return metas


def run_sequence(tracker, video, query_points, query_mode, flow_cache=None, device=None):
    """One sequence, one query mode: ``video`` (n_frames, H, W, 3) uint8 BGR, ``query_points``
    (n_queries, 3) as (t, y, x) in video pixels.  Returns {'tracks': (1, n, T, 2) xy on the
    256 x 256 raster, 'occluded': (1, n, T) occlusion scores} -- the reference's per-sequence
    tracklet pickle.  Occlusion scores are left soft, as the reference stores them."""
    if query_mode not in ("first", "strided"):
        raise ValueError("Unknown query mode " + query_mode)
    query_points = np.asarray(query_points).astype(np.int64)
    n_frames, H, W = len(video), video[0].shape[0], video[0].shape[1]
    n_q = query_points.shape[0]
    pred_tracks = np.zeros((n_q, n_frames, 2))
    pred_occluded = np.zeros((n_q, n_frames))
    directions = ("forward", "backward") if query_mode == "strided" else ("forward",)
    for start_frame in np.unique(query_points[:, 0]):
        sel = query_points[:, 0] == start_frame
        queries_xy = torch.from_numpy(query_points[sel, 1:][:, ::-1].copy())
        if device is not None:
            queries_xy = queries_xy.to(device)
        for direction in directions:
            metas = track_sequence(tracker, video, int(start_frame), direction=direction, flow_cache=flow_cache)
            for frame_i, meta in metas.items():
                coords, occl = convert_to_point_tracking(meta.result, queries_xy)
                pred_tracks[sel, frame_i] = coords
                pred_occluded[sel, frame_i] = occl
    pred_tracks *= np.array([256.0 / W, 256.0 / H])
    return {"tracks": pred_tracks[None], "occluded": pred_occluded[None]}


def evaluate(outputs, gt, query_mode, occlusion_threshold=0.5):
    """Metrics of one ``run_sequence`` output against a ``sample_queries_*`` dict whose points are
    in video pixels of an (H, W) video: everything is rescaled to 256 x 256 first
    (``MFT/runners/eval_MFT_tapvid.py``: scores above the threshold count as occluded)."""
    H, W = gt["video"].shape[2:4]
    scale = np.array([256.0 / W, 256.0 / H])
    q = gt["query_points"].astype(np.float64).copy()
    q[..., 1] *= scale[1]
    q[..., 2] *= scale[0]
    return compute_tapvid_metrics(q, gt["occluded"].astype(bool), gt["target_points"] * scale,
                                  outputs["occluded"] > occlusion_threshold, outputs["tracks"], query_mode)


def synthetic_sequence(video, n_tracks=32, seed=0):
    """TAP-Vid-shaped ground truth on a ``SyntheticVideo``: integer background points of random
    frames -> (target_occluded (n, T) bool, target_points (n, T, 2) xy, frames (T, H, W, 3))."""
    rng = np.random.Generator(np.random.PCG64([seed, 0x7A9]))
    T = len(video)
    occ, pts = [], []
    while len(occ) < n_tracks:
        t = int(rng.integers(0, T))
        p = np.array([rng.integers(0, video.W), rng.integers(0, video.H)], np.float64)
        x0, y0, side = video.occluder(t)
        if x0 <= p[0] < x0 + side and y0 <= p[1] < y0 + side:
            continue                                   # background points only
        tr, oc = video.ground_truth_tracks(p[None], t)
        occ.append(oc[0])
        pts.append(tr[0])
    frames = np.stack([video[i] for i in range(T)])
    return np.stack(occ), np.stack(pts), frames
