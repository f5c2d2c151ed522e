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


# ---------------------------------------------------------------------------
# dataset reader (create_tapvid_dataset / parse_scale_WH / Kinetics shards) vs the reference's own reader run on the
# fixture pickles (tests/golden/tapvid_dataset.npz, tools/make_goldens.py tapvid_dataset)
# ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def gd(golden_dir):
    return np.load(golden_dir / "tapvid_dataset.npz")


def test_parse_scale_WH_matches_reference(gd):
    shapes = [{"N_frames": 8, "H": 40, "W": 56, "C": 3}, {"N_frames": 3, "H": 480, "W": 854, "C": 3}]
    n = 0
    for j, fs in enumerate(shapes):
        for sc in ("fullres", "256x256", "x1080", "512x", "256x256_x480", "fullres_300x", "x7_9x_fullres"):
            got = tapvid.parse_scale_WH(sc, fs)
            want = gd[f"parse|{j}|{sc}"]
            assert [[d["N_frames"], d["H"], d["W"], d["C"]] for d in got] == want.tolist(), sc
            n += 1
    assert n == 14
    assert tapvid.parse_scale_WH("256x256_512x512", shapes[1])[-1]["W"] == 512      # the BASELINE config 3 string
    with pytest.raises(AssertionError):
        tapvid.parse_scale_WH("x", shapes[0])


@pytest.mark.parametrize("form,fname", [("dict", "tapvid_davis_like.pkl"), ("kin", "tapvid_kinetics_like_0007.pkl")])
def test_create_tapvid_dataset_matches_reference(gd, golden_dir, form, fname):
    """Every scaling kind x both pickle forms: names, N_sequences, the resized video (bitwise; by checksum for the big ones), and
    the sampled query / target / occlusion tables of both query modes -- all equal to what the reference's reader returned."""
    import zlib
    seen = 0
    for sc in gi.TAPVID_SCALINGS:
        arg = {"default": None, "false": False}.get(sc, sc)
        for fake in (False, True):
            if fake and sc not in ("256x256_512x512", "x30"):
                continue
            els = list(tapvid.create_tapvid_dataset(golden_dir / fname, ["first", "strided"], arg, fake_video=fake))
            assert len(els) == 2
            for i, el in enumerate(els):
                key = f"{form}|{sc}|{'fake' if fake else 'real'}|{i}"
                assert el["video_name"] == str(gd[key + "|name"]) and el["N_sequences"] == int(gd[key + "|N"])
                assert sorted(el["data"]) == ["first", "strided"]
                for mode in ("first", "strided"):
                    g = el["data"][mode]
                    assert list(g["video"].shape) == gd[f"{key}|{mode}|video_shape"].tolist(), key
                    assert zlib.crc32(np.ascontiguousarray(g["video"]).tobytes()) == int(gd[f"{key}|{mode}|video_crc"]), key
                    for k in ("query_points", "target_points", "occluded", "trackgroup"):
                        want = gd[f"{key}|{mode}|{k}"]
                        assert g[k].shape == want.shape and np.array_equal(np.asarray(g[k], want.dtype), want), (key, mode, k)
                if key + "|video" in gd.files:
                    assert np.array_equal(el["data"]["first"]["video"], gd[key + "|video"])
                seen += 1
    assert seen == 2 * (len(gi.TAPVID_SCALINGS) + 2)
    # the BASELINE config-3 string: tracked at 512 x 512 on frames resampled 40x56 -> 256x256 -> 512x512, points at 512 scale
    el = next(iter(tapvid.create_tapvid_dataset(golden_dir / fname, ["first"], "256x256_512x512")))
    assert el["data"]["first"]["video"].shape[2:4] == (512, 512) and el["data"]["first"]["target_points"].max() > 256


def test_create_tapvid_dataset_rejects_lazy(golden_dir):
    with pytest.raises(ValueError):
        next(tapvid.create_tapvid_dataset(golden_dir / "tapvid_davis_like.pkl", ["first"], lazy_video=True))
    frames = tapvid.resize_video(np.zeros((3, 8, 8, 3), np.uint8), (4, 6), lazy_video=True)
    assert len(frames) == 3 and frames[1]().shape == (4, 6, 3)
    assert tapvid.resize_video(np.zeros((3, 8, 8, 3), np.uint8), (4, 6), fake_video=True).shape == (3, 4, 6, 3)


# ---------------------------------------------------------------------------
# dataset runner + evaluation (run_MFT_tapvid.py:100-237, eval_MFT_tapvid.py:69-133) on the fixture pickle, stub flows
# ---------------------------------------------------------------------------
class HashFlower:
    """Stub flow plugin for arbitrary frames: recognises a frame by its bytes (the frames of the sequences it was given)."""

    def __init__(self):
        self.ids, self.calls = {}, []

    def learn(self, video, tag):
        import zlib
        for i, f in enumerate(video):
            self.ids[zlib.crc32(np.ascontiguousarray(f).tobytes())] = (tag, i)

    def compute_flow(self, src_img, dst_img, mode="flow", init_flow=None, **kw):
        import zlib
        (ta, l), (tb, r) = (self.ids[zlib.crc32(np.ascontiguousarray(np.asarray(x)).tobytes())] for x in (src_img, dst_img))
        assert ta == tb
        self.calls.append((ta, l, r))
        H, W = np.asarray(src_img).shape[:2]
        flow, occl, sigma = gi.stub_flowou(l, r, H, W)
        return torch.from_numpy(flow * 0.25), {"occlusion": torch.from_numpy(occl), "sigma": torch.from_numpy(sigma), "debug": None}


def _dataset_conf(golden_dir, scaling):
    from mft_amd.config import Config
    c = Config()
    c.pickles = [golden_dir / "tapvid_davis_like.pkl"]
    c.scaling = scaling
    c.name = "fixture-" + scaling
    return c


def test_run_dataset_and_evaluate(golden_dir, tmp_path):
    import pickle
    from mft_amd.io import FlowCache
    scaling = "32x24_64x48"                       # track at 64 x 48 on twice-resampled frames, score at 256 x 256
    dconf = _dataset_conf(golden_dir, scaling)
    fl = HashFlower()
    for el in tapvid.create_tapvid_dataset(dconf.pickles[0], ["first"], scaling):
        fl.learn(el["data"]["first"]["video"][0][..., ::-1], el["video_name"])
    tr = make_tracker(fl, deltas=(np.inf, 1, 2, 4))
    tr.C.name = "stubtracker"
    tr.C.flow_config.name = "stubflow"
    tr.C.tracker_class = None
    caches = []

    def factory(d, ram, gpu):
        caches.append(FlowCache(d, max_RAM_MB=ram, max_GPU_RAM_MB=gpu, device="cpu"))
        return caches[-1]
    export, cache_root = tmp_path / "export", tmp_path / "cache"
    done = tapvid.run_dataset(dconf, [tr.C], export, cache_root, mode="both", tracker=tr, device=None, cache_factory=factory,
                              write_flow=False)
    assert [(d["sequence"], d["mode"], d["skipped"]) for d in done] == [
        ("seq-wide", "first", False), ("seq-wide", "strided", False), ("seq-square", "first", False), ("seq-square", "strided", False)]
    assert len(caches) == 2 and not any((cache_root / dconf.name / "stubflow").glob("*"))      # one cache per sequence, removed after
    # the files hold what run_sequence returns for the same queries (a fresh tracker, a fresh cache)
    els = {el["video_name"]: el for el in tapvid.create_tapvid_dataset(dconf.pickles[0], ["first", "strided"], scaling)}
    for d in done:
        with open(d["path"], "rb") as f:
            out = pickle.load(f)
        assert d["path"] == export / "stubtracker" / "results" / f"{d['sequence']}-{d['mode']}.pklz"
        el = els[d["sequence"]]["data"][d["mode"]]
        video = np.ascontiguousarray(el["video"][0][..., ::-1])
        assert video.shape[1:3] == (48, 64)
        want = tapvid.run_sequence(make_tracker(fl, deltas=(np.inf, 1, 2, 4)), video, el["query_points"][0], d["mode"],
                                   flow_cache=FlowCache(None, device="cpu"))
        assert np.array_equal(out["tracks"], want["tracks"]) and np.array_equal(out["occluded"], want["occluded"])
        assert out["tracks"].shape == el["target_points"].shape
        assert out["tracks"][..., 0].max() <= 256 * 1.5           # on the 256 raster, whatever the tracking size
    # cont: nothing is recomputed; without it everything is
    n_calls = len(fl.calls)
    again = tapvid.run_dataset(dconf, [tr.C], export, cache_root, mode="both", cont=True, tracker=tr, device=None, cache_factory=factory)
    assert all(d["skipped"] for d in again) and len(fl.calls) == n_calls
    sub = tapvid.run_dataset(dconf, [tr.C], export, cache_root, mode="first", seqs=["seq-square"], tracker=tr, device=None,
                             cache_factory=factory)
    assert [(d["sequence"], d["mode"]) for d in sub] == [("seq-square", "first")] and len(fl.calls) > n_calls
    # evaluation: one row per sequence, the reference's keys, written as DataFrames
    m = tapvid.evaluate_dataset(dconf, [tr.C], export, mode="both")
    assert sorted(m) == ["first", "strided"]
    rows = m["strided"]["stubtracker"]
    assert [r["seq"] for r in rows] == ["seq-wide", "seq-square"] and 0.0 <= rows[0]["average_jaccard"] <= 1.0
    import pandas as pd
    df = pd.read_pickle(export / "stubtracker" / "eval" / "tapvid-eval-strided.pklz")
    assert list(df["seq"]) == ["seq-wide", "seq-square"] and "average_pts_within_thresh" in df.columns
    assert (export / "stubtracker" / "eval" / "tapvid-eval.pklz").exists()
    # ground truth as the prediction scores 1: the scale chain of the evaluation is consistent with the runner's
    for d in done:
        el = els[d["sequence"]]["data"][d["mode"]]
        H, W = el["video"].shape[2:4]
        with open(d["path"], "wb") as f:
            pickle.dump({"tracks": el["target_points"] * np.array([256.0 / W, 256.0 / H]), "occluded": el["occluded"].astype(np.float64)}, f)
    m = tapvid.evaluate_dataset(dconf, [tr.C], export, mode="both", write=False)
    assert all(r["average_jaccard"] == 1.0 and r["occlusion_accuracy"] == 1.0 for mode in m for r in m[mode]["stubtracker"])


@pytest.mark.gpu
def test_run_dataset_write_flow_exports_template_results(golden_dir, tmp_path):
    """(gpu: the X16 writer quantises on the device)"""
    from mft_amd.io import FlowCache
    from mft_amd.results import FlowOUTrackingResult
    dconf = _dataset_conf(golden_dir, "fullres")
    fl = HashFlower()
    for el in tapvid.create_tapvid_dataset(dconf.pickles[0], ["first"], "fullres"):
        # (not `False`: like the reference, a non-string size is rebound by the first sequence and kept for the rest)
        fl.learn(el["data"]["first"]["video"][0][..., ::-1], el["video_name"])
    tr = make_tracker(fl, deltas=(np.inf, 1, 2))
    tr.C.name, tr.C.flow_config.name, tr.C.tracker_class = "t", "f", None
    factory = lambda d, ram, gpu: FlowCache(d, max_RAM_MB=ram, max_GPU_RAM_MB=gpu, device="cpu")    # noqa: E731
    try:
        tapvid.run_dataset(dconf, [tr.C], tmp_path / "e", tmp_path / "c", mode="first", tracker=tr, device=None,
                           cache_factory=factory, write_flow=True, seqs=["seq-square"])
    except Exception as e:                      # (only if no track of the fixture is visible in frame 0)
        assert "0 is not in 'start_frames'" in str(e)
        return
    files = sorted((tmp_path / "e" / "t" / "flowous" / "seq-square").glob("0--*.flowouX16.pkl"))
    assert len(files) == 8
    r = FlowOUTrackingResult.read(files[3])
    assert r.flow.shape == (2, 36, 36)
