"""TAP-Vid protocol: samplers and metrics against outputs of the
reference's own functions on seeded inputs (tests/golden/tapvid.npz, made by
tools/make_goldens.py tapvid), and the per-sequence runner on a stub flow."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
from mft_amd import tapvid
from test_host_logic import StubFlower, make_tracker


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(golden_dir / "tapvid.npz")


@pytest.mark.parametrize("mode", ["first", "strided"])
def test_query_sampling_matches_reference(g, mode):
    d = gi.tapvid_inputs()
    for v in range(2):
        occ, pts = d["gt_occluded"][v], d["gt_tracks"][v]
        s = (tapvid.sample_queries_first(occ, pts, d["frames"]) if mode == "first"
             else tapvid.sample_queries_strided(occ, pts, d["frames"], query_stride=5))
        assert s["video"].shape == (1,) + d["frames"].shape
        for k in ("query_points", "target_points", "occluded", "trackgroup"):
            want = g[f"{mode}_{v}_{k}"]
            assert s[k].shape == want.shape, k
            assert np.array_equal(np.asarray(s[k], want.dtype), want), k


@pytest.mark.parametrize("mode", ["first", "strided"])
def test_metrics_match_reference(g, mode):
    d = gi.tapvid_inputs()
    for v in range(2):
        q, tp, oc, tg = (g[f"{mode}_{v}_{k}"] for k in ("query_points", "target_points", "occluded", "trackgroup"))
        m = tapvid.compute_tapvid_metrics(q, oc, tp, d["pred_occluded"][v][tg[0]][None],
                                          d["pred_tracks"][v][tg[0]][None], mode)
        keys = [k[len(f"{mode}_{v}_metric_"):] for k in g.files if k.startswith(f"{mode}_{v}_metric_")]
        assert sorted(m) == sorted(keys)
        for k in keys:
            assert np.array_equal(m[k], g[f"{mode}_{v}_metric_{k}"]), k
    # batched call
    q = np.zeros((2, d["gt_tracks"].shape[1], 3))
    occ = d["gt_occluded"].copy()
    occ[:, :, 0] = False
    m = tapvid.compute_tapvid_metrics(q, occ, d["gt_tracks"], d["pred_occluded"], d["pred_tracks"], mode)
    for k, val in m.items():
        assert np.array_equal(val, g[f"{mode}_batch_metric_{k}"]), k
    assert m["average_jaccard"].shape == (2,)


def test_metrics_reject_unknown_mode():
    z = np.zeros((1, 2, 3))
    with pytest.raises(ValueError):
        tapvid.compute_tapvid_metrics(z, np.zeros((1, 2, 3), bool), np.zeros((1, 2, 3, 2)),
                                      np.zeros((1, 2, 3), bool), np.zeros((1, 2, 3, 2)), "last")


def test_perfect_prediction_scores_one():
    d = gi.tapvid_inputs()
    s = tapvid.sample_queries_strided(d["gt_occluded"][0], d["gt_tracks"][0], d["frames"])
    m = tapvid.compute_tapvid_metrics(s["query_points"], s["occluded"], s["target_points"], s["occluded"],
                                      s["target_points"], "strided")
    assert m["occlusion_accuracy"][0] == 1.0 and m["average_jaccard"][0] == 1.0 and m["average_pts_within_thresh"][0] == 1.0


def test_runner_strided_uses_one_cache_and_both_directions():
    """Queries on frames 0 and 5 of a 10-frame stub video: every frame of every query row is
    filled (backward run covers the frames before the query), flow pairs requested by the second
    start frame that the first already produced come from the shared cache."""
    from mft_amd.io import FlowCache
    n = 10
    video = [gi.id_image(i) for i in range(n)]
    H, W = video[0].shape[:2]
    fl = StubFlower()
    tr = make_tracker(fl)
    cache = FlowCache(None, device="cpu")
    qp = np.array([[0, 10, 12], [0, 30, 40], [5, 20, 22]])          # (t, y, x)
    out = tapvid.run_sequence(tr, video, qp, "strided", flow_cache=cache)
    assert out["tracks"].shape == (1, 3, n, 2) and out["occluded"].shape == (1, 3, n)
    # the query frame itself: identity result -> the query point, on the 256 raster
    assert np.allclose(out["tracks"][0, 0, 0], [12 * 256.0 / W, 10 * 256.0 / H])
    assert np.allclose(out["tracks"][0, 2, 5], [22 * 256.0 / W, 20 * 256.0 / H])
    # frames before the query frame were reached by the backward run
    assert not np.allclose(out["tracks"][0, 2, 0], 0.0)
    # every pair is computed once across the four runs -- except the delta = inf pairs of the second
    # start frame that coincide with a finite delta of the first run: infinity pairs bypass the
    # cache unless C.cache_delta_infinity (MFT/MFT.py:99)
    import collections
    dup = sorted(k for k, c in collections.Counter(fl.calls).items() if c > 1)
    assert dup == [(5, 6), (5, 7), (5, 9)]
    # with infinity pairs cached too, nothing is ever computed twice and a second pass computes nothing
    fl2 = StubFlower()
    tr2 = make_tracker(fl2, cache_delta_infinity=True)
    cache2 = FlowCache(None, device="cpu")
    out2 = tapvid.run_sequence(tr2, video, qp, "strided", flow_cache=cache2)
    assert len(fl2.calls) == len(set(fl2.calls)) == len(set(fl.calls))
    n_first = len(fl2.calls)
    out3 = tapvid.run_sequence(tr2, video, qp, "strided", flow_cache=cache2)
    assert len(fl2.calls) == n_first
    assert np.array_equal(out2["tracks"], out["tracks"]) and np.array_equal(out3["tracks"], out["tracks"])
    # 'first' mode runs forward only
    out1 = tapvid.run_sequence(tr, video, qp, "first", flow_cache=cache)
    assert np.all(out1["tracks"][0, 2, :5] == 0.0)


def test_track_sequence_meta_fields():
    video = [gi.id_image(i) for i in range(6)]
    tr = make_tracker(StubFlower())
    metas = tapvid.track_sequence(tr, video, 4, direction="backward")
    assert sorted(metas) == [0, 1, 2, 3, 4]
    assert all(m.backward and m.frame_i == i for i, m in metas.items())
    metas = tapvid.track_sequence(tr, video, 2, direction="forward")
    assert sorted(metas) == [2, 3, 4, 5] and not metas[3].backward


def test_synthetic_ground_truth_is_consistent():
    """A background point's track, re-derived from any other frame of it, is the same track."""
    from mft_amd.synth import SyntheticVideo
    vid = SyntheticVideo(96, 128, n_frames=12, seed=3)
    occ, pts, frames = tapvid.synthetic_sequence(vid, n_tracks=6, seed=1)
    assert occ.shape == (6, 12) and pts.shape == (6, 12, 2) and frames.shape == (12, 96, 128, 3)
    tr2, oc2 = vid.ground_truth_tracks(pts[:, 7], 7)
    assert np.allclose(tr2, pts, atol=1e-6) and np.array_equal(oc2, occ)
    s = tapvid.sample_queries_first(occ, pts, frames)
    m = tapvid.evaluate({"tracks": s["target_points"] * np.array([256 / 128, 256 / 96]),
                         "occluded": s["occluded"].astype(np.float64)}, s, "first")
    assert m["average_jaccard"][0] == 1.0
