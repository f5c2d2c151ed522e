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

  * ``MFT/evaluation/tapvid_eval_stuff.py:612-672`` -- ``create_tapvid_dataset``: the pickled TAP-Vid
    shards (a dict of sequences for DAVIS / RGB-stacking, a list of JPEG-encoded sequences for the
    Kinetics shards, ``:528-549``), the chain of rescalings a ``scaling`` string such as
    ``'256x256_512x512'`` asks for (``MFT/utils/misc.py:65-92``: resize to 256 x 256, THEN to 512 x 512 --
    the tracker runs at 512 x 512 on twice-resampled frames, scores are taken on the 256 x 256 raster) and
    ``resize_video`` (``:61-80``).
  * ``MFT/runners/run_MFT_tapvid.py:100-237`` / ``eval_MFT_tapvid.py:69-133`` -- the per-dataset runner
    (``run_dataset``: one flow cache per sequence, ``<seq>-<mode>.pklz`` tracklet pickles, ``cont``
    skipping, flowou export of the frame-0 template) and its evaluation (``evaluate_dataset``).

No dataset is reachable from the build environment: ``synthetic_sequence`` / ``synthetic_pickle`` make
TAP-Vid-shaped ground truth (and a TAP-Vid-shaped pickle) from the seeded synthetic video instead.
"""
from __future__ import annotations

import io as _pyio
import pickle
import shutil
from pathlib import Path

import numpy as np
import torch

from .point_tracking import convert_to_point_tracking

THRESHOLDS = (1, 2, 4, 8, 16)
TRAIN_SIZE = (24, 256, 256, 3)          # tapnet's training raster: the default target of create_tapvid_dataset


# ---------------------------------------------------------------------------
# dataset reader
# ---------------------------------------------------------------------------
def parse_scale_WH(scale_WH, frames_shape):
    """'fullres' | 'WxH' | 'Wx' | 'xH', several joined by '_' = a SEQUENCE of rescalings -> list of shape dicts (the keys
    of ``frames_shape`` with 'W' / 'H' replaced; a missing side keeps the ORIGINAL aspect ratio, rounded).
    ``MFT/utils/misc.py:65-92``."""
    if scale_WH == "fullres":
        return [frames_shape]
    out = []
    for part in scale_WH.split("_"):
        if part == "fullres":
            out.append(frames_shape)
            continue
        W_str, H_str = part.split("x")
        W = int(W_str) if W_str != "" else None
        H = int(H_str) if H_str != "" else None
        assert W is not None or H is not None, "at least one dimmension has to be set"
        shape = dict(frames_shape.items())
        shape["W"] = W if W is not None else int(round(frames_shape["W"] * (H / frames_shape["H"])))
        shape["H"] = H if H is not None else int(round(frames_shape["H"] * (W / frames_shape["W"])))
        out.append(shape)
    return out


def resize_frame(frame, output_size):
    """One uint8 frame [H, W, C] -> [output_size[0], output_size[1], C] with PIL's Lanczos filter: what
    ``mediapy.resize_video`` (the reference's resizer, ``tapvid_eval_stuff.py:80``) does for uint8 RGB frames."""
    from PIL import Image
    frame = np.asarray(frame)
    assert frame.dtype == np.uint8 and frame.ndim == 3
    H, W = int(output_size[0]), int(output_size[1])
    return np.asarray(Image.fromarray(frame).resize((W, H), resample=Image.Resampling.LANCZOS), dtype=np.uint8)


def resize_video(video, output_size, fake_video=False, lazy_video=False):
    """(N, H, W, C) uint8 -> (N, output_size[0], output_size[1], C).  ``fake_video``: zeros of the right shape (the evaluation
    only needs the shape, ``eval_MFT_tapvid.py:82``).  ``lazy_video``: a list of zero-argument callables, one per frame
    (the reference's version of this branch returns callables that return None, ``tapvid_eval_stuff.py:64-69``)."""
    video = np.asarray(video)
    N, _, _, C = video.shape
    if lazy_video:
        return [(lambda i=i: resize_frame(video[i], output_size)) for i in range(N)]
    if fake_video:
        return np.zeros((N, int(output_size[0]), int(output_size[1]), C), dtype=video.dtype)
    return np.stack([resize_frame(video[i], output_size) for i in range(N)]) if N else \
        np.zeros((0, int(output_size[0]), int(output_size[1]), C), dtype=video.dtype)


def load_kinetics_video(data):
    """A Kinetics shard entry: ``data['video']`` is a list of JPEG byte strings -> (N, H, W, 3) uint8 RGB, in place
    (``tapvid_eval_stuff.py:528-549``)."""
    from PIL import Image
    frames = []
    for byte_string in data["video"]:
        img = np.asarray(Image.open(_pyio.BytesIO(byte_string)))
        assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3
        frames.append(img)
    data["video"] = np.array(frames)
    return data


def create_tapvid_dataset(pickle_path, query_modes, train_size=None, fake_video=False, lazy_video=False):
    """Generator over the sequences of one TAP-Vid pickle (``tapvid_eval_stuff.py:612-672``):
    ``{'data': {mode: sample_queries_<mode>(...)}, 'video_name': str, 'N_sequences': int}``.

    pickle: ``{name: {'video' (N, H, W, 3) uint8 RGB, 'points' (n, N, 2) xy in [0, 1], 'occluded' (n, N) bool}}`` or a LIST
    of such dicts with JPEG-encoded frames (a Kinetics shard; sequences are then named ``kin-<shard>-<i:04d>``).
    train_size: None -> 256 x 256 (tapnet's raster); False -> the video's own size; a ``parse_scale_WH`` string ->
    the frames go through EVERY rescaling of the string in turn and the points are scaled to the LAST one; a tuple
    (_, H, W, _) -> that size.  Like the reference a size, once set, is kept for the remaining sequences of the pickle
    (``train_size`` is rebound inside the loop) and the points of the loaded pickle are scaled in place."""
    if lazy_video:
        # the reference's lazy branch yields a list and then trips over `frames.shape` (tapvid_eval_stuff.py:650)
        raise ValueError("create_tapvid_dataset: lazy_video is not usable (nor is it in the reference)")
    if train_size is None:
        train_size = TRAIN_SIZE
    size_str = train_size if isinstance(train_size, str) else None
    with open(pickle_path, "rb") as f:
        dataset = pickle.load(f)
    if isinstance(dataset, list):
        shard = Path(pickle_path).stem
        dataset = {f"kin-{shard}-{i:04d}": load_kinetics_video(d) for i, d in enumerate(dataset)}
    n_sequences = len(dataset)
    for name in dataset:
        frames = np.asarray(dataset[name]["video"])
        N, H, W, C = frames.shape
        shape = {"N_frames": N, "H": H, "W": W, "C": C}
        if size_str is not None:
            for scaled in parse_scale_WH(size_str, shape):
                train_size = (1, scaled["H"], scaled["W"], C)
                frames = resize_video(frames, train_size[1:3], fake_video=fake_video)
        elif train_size is False:
            train_size = (1, H, W, C)
            frames = resize_video(frames, train_size[1:3], fake_video=fake_video)
        else:
            frames = resize_video(frames, train_size[1:3], fake_video=fake_video)
        assert frames.ndim == 4 and frames.shape[0] == N and frames.shape[3] == C
        points = dataset[name]["points"]
        occluded = dataset[name]["occluded"]
        points *= np.array([train_size[2], train_size[1]])
        converted = {}
        if "strided" in query_modes:
            converted["strided"] = sample_queries_strided(occluded, points, frames)
        if "first" in query_modes:
            converted["first"] = sample_queries_first(occluded, points, frames)
        yield {"data": converted, "video_name": name, "N_sequences": n_sequences}


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


def run_sequence(tracker, video, query_points, query_mode, flow_cache=None, device=None, on_run=None, debug=False):
    """One sequence, one query mode: ``video`` (n_frames, H, W, 3) uint8 BGR, ``query_points``
    (n_queries, 3) as (t, y, x) in video pixels.  Returns {'tracks': (1, n, T, 2) xy on the
    256 x 256 raster, 'occluded': (1, n, T) occlusion scores} -- the reference's per-sequence
    tracklet pickle.  Occlusion scores are left soft, as the reference stores them.
    ``on_run(start_frame, direction, metas)``: called after every tracker run (exports)."""
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
            metas = track_sequence(tracker, video, int(start_frame), direction=direction, debug=debug, flow_cache=flow_cache)
            for frame_i, meta in metas.items():
                coords, occl = convert_to_point_tracking(meta.result, queries_xy)
                pred_tracks[sel, frame_i] = coords
                pred_occluded[sel, frame_i] = occl
            if on_run is not None:
                on_run(int(start_frame), direction, metas)
    pred_tracks *= np.array([256.0 / W, 256.0 / H])
    return {"tracks": pred_tracks[None], "occluded": pred_occluded[None]}


def _query_modes(mode):
    if mode not in ("first", "strided", "both"):
        raise ValueError("Unknown query mode " + str(mode))
    return ["first", "strided"] if mode == "both" else [mode]


def validate_configs(configs):
    """All tracker configs of one run share the tracker class and the flow config: ONE tracker object is built and its ``C`` is
    swapped between runs (``run_MFT_tapvid.py:89-96, 151, 297-300``)."""
    assert all(c.tracker_class == configs[0].tracker_class for c in configs)
    assert all(c.flow_config == configs[0].flow_config for c in configs)


def result_path(export, tracker_name, sequence_name, query_mode):
    return Path(export) / tracker_name / "results" / f"{sequence_name}-{query_mode}.pklz"


def run_dataset(dataset_conf, configs, export, cache_root, mode="both", cont=False, seqs=None, write_flow=False,
                ram_cache_limit=30, gpu_cache_limit=5, tracker=None, device="cuda", debug=False, cache_factory=None):
    """The TAP-Vid run of ``MFT/runners/run_MFT_tapvid.py:85-247``: every sequence of every pickle of ``dataset_conf``
    (``.pickles``, ``.scaling``, ``.name``) is tracked for every query mode and tracker config, all runs of a sequence
    sharing one flow cache (``cache_root/<dataset>/<flow name>/<sequence>``, emptied before and removed after), and the
    tracklets are pickled to ``export/<tracker name>/results/<sequence>-<mode>.pklz`` as {'tracks' (1, n, T, 2) on the
    256 x 256 raster, 'occluded' (1, n, T) scores}.  ``cont``: existing result files are skipped.  ``write_flow``: the
    frame-0 template's results of the 'first' run are written as ``flowous/<sequence>/0--<i>.flowouX16.pkl``.
    ``cache_factory(dir, max_RAM_MB, max_GPU_RAM_MB)`` defaults to ``mft_amd.io.FlowCache`` (HBM tier first: on an MI355X
    ``gpu_cache_limit`` can be raised to hundreds of GB).  -> list of {'sequence', 'mode', 'tracker', 'path', 'skipped'}."""
    from .io import FlowCache
    configs = list(configs)
    validate_configs(configs)
    if tracker is None:
        tracker = configs[0].tracker_class(configs[0])
    export, cache_root = Path(export), Path(cache_root)
    for c in configs:
        (export / c.name / "results").mkdir(parents=True, exist_ok=True)
    modes = _query_modes(mode)
    make_cache = cache_factory or (lambda d, ram, gpu: FlowCache(d, max_RAM_MB=ram, max_GPU_RAM_MB=gpu, device=device))
    done = []
    for pickle_path in dataset_conf.pickles:
        for seq in create_tapvid_dataset(pickle_path, modes, dataset_conf.scaling):
            name = seq["video_name"]
            if seqs is not None and name not in seqs:
                continue
            video = seq["data"][modes[0]]["video"][0]                   # every mode carries the same video
            video = np.ascontiguousarray(video[:, :, :, ::-1])           # RGB -> BGR, what the tracker takes
            assert video.dtype == np.uint8 and video.shape[3] == 3
            flow_name = configs[0].flow_config.name
            assert flow_name
            cache_dir = cache_root / str(dataset_conf.name) / str(flow_name) / name
            shutil.rmtree(cache_dir, ignore_errors=True)
            cache_dir.mkdir(parents=True, exist_ok=True)
            cache = make_cache(cache_dir, ram_cache_limit * 1e3, gpu_cache_limit * 1e3)
            for query_mode in modes:
                query_points = np.asarray(seq["data"][query_mode]["query_points"])[0].astype(np.int64)
                if query_mode == "first" and write_flow and 0 not in np.unique(query_points[:, 0]):
                    raise Exception("Trying to export flowous from first frame, but 0 is not in 'start_frames'" +
                                    f"in {name} in {Path(pickle_path).stem}")
                for cfg in configs:
                    tracker.C = cfg
                    path = result_path(export, cfg.name, name, query_mode)
                    if cont and path.exists():
                        done.append(dict(sequence=name, mode=query_mode, tracker=cfg.name, path=path, skipped=True))
                        continue

                    def on_run(start_frame, direction, metas, _cfg=cfg, _mode=query_mode):
                        if start_frame == 0 and _mode == "first" and write_flow and direction == "forward":
                            flowou_dir = export / _cfg.name / "flowous" / name
                            flowou_dir.mkdir(parents=True, exist_ok=True)
                            for frame_i, meta in metas.items():
                                meta.result.write(flowou_dir / f"0--{frame_i}.flowouX16.pkl")
                    out = run_sequence(tracker, video, query_points, query_mode, flow_cache=cache, device=device, on_run=on_run,
                                       debug=debug)
                    assert out["tracks"].shape[0] == 1 and out["tracks"].shape[3] == 2 and out["tracks"].ndim == 4
                    with open(path, "wb") as f:
                        pickle.dump(out, f)
                    done.append(dict(sequence=name, mode=query_mode, tracker=cfg.name, path=path, skipped=False))
            shutil.rmtree(cache_dir, ignore_errors=True)
            cache.clear()
    return done


def evaluate_dataset(dataset_conf, configs, export, mode="both", write=True):
    """``MFT/runners/eval_MFT_tapvid.py:69-133``: the tracklet pickles of ``run_dataset`` against the ground truth of the same
    pickles (frames are not needed: ``fake_video``), both on the 256 x 256 raster, occlusion scores thresholded at 0.5.
    -> {mode: {tracker name: [metrics dict per sequence, + 'seq']}}; with ``write`` each list also goes to
    ``export/<tracker>/eval/tapvid-eval[-strided].pklz`` as a pandas DataFrame, like the reference."""
    modes = _query_modes(mode)
    export = Path(export)
    all_metrics = {m: {c.name: [] for c in configs} for m in modes}
    for pickle_path in dataset_conf.pickles:
        for seq in create_tapvid_dataset(pickle_path, modes, dataset_conf.scaling, fake_video=True):
            name = seq["video_name"]
            H, W = seq["data"][modes[0]]["video"].shape[2:4]
            scale = np.array([256.0 / W, 256.0 / H])
            for query_mode in modes:
                gt = seq["data"][query_mode]
                query_points = np.asarray(gt["query_points"])[0].astype(np.int64)[None]
                gt_tracks = gt["target_points"] * scale
                gt_occluded = gt["occluded"]
                for cfg in configs:
                    path = result_path(export, cfg.name, name, query_mode)
                    if not path.exists():
                        continue                               # (a subset run: --seq)
                    with open(path, "rb") as f:
                        out = pickle.load(f)
                    pred_occluded = np.float32(out["occluded"] > 0.5)
                    assert out["tracks"].shape == gt_tracks.shape and pred_occluded.shape == gt_occluded.shape
                    m = compute_tapvid_metrics(query_points, gt_occluded, gt_tracks, pred_occluded, out["tracks"], query_mode)
                    assert all(v.shape == (1,) for v in m.values())
                    m = {k: v[0] for k, v in m.items()}
                    m["seq"] = name
                    all_metrics[query_mode][cfg.name].append(m)
    if write:
        import pandas as pd
        for cfg in configs:
            eval_dir = export / cfg.name / "eval"
            eval_dir.mkdir(parents=True, exist_ok=True)
            for query_mode in modes:
                rows = dict(enumerate(all_metrics[query_mode][cfg.name]))
                pd.DataFrame.from_dict(rows, orient="index").to_pickle(
                    eval_dir / ("tapvid-eval-strided.pklz" if query_mode == "strided" else "tapvid-eval.pklz"))
    return all_metrics


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


def synthetic_pickle(path, videos, n_tracks=24, seed=0):
    """Write a TAP-Vid-shaped pickle (dict form: RGB frames, points normalised to [0, 1], occlusion flags) for
    ``{name: SyntheticVideo}`` -- what ``create_tapvid_dataset`` reads; the analytic ground truth of ``synthetic_sequence``."""
    data = {}
    for i, (name, video) in enumerate(videos.items()):
        occ, pts, frames = synthetic_sequence(video, n_tracks=n_tracks, seed=seed + i)
        data[name] = {"video": np.ascontiguousarray(frames[..., ::-1]),          # BGR -> RGB
                      "points": pts / np.array([float(video.W), float(video.H)]),
                      "occluded": occ}
    with open(path, "wb") as f:
        pickle.dump(data, f, protocol=4)
    return data


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
