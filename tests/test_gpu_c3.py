"""BASELINE.json configs[2] (C3): the TAP-Vid-DAVIS protocol with the REAL engine -- dataset pickle -> reader with the
'256x256_512x512' scaling -> per-sequence runner ('first' + 'strided', forward + backward, one flow cache) -> tracklet
pickles -> evaluation -- on the HIP tracker, against the same protocol driven through the CPU oracle tracker
(VERDICT round 4, item 1).  The dataset itself is not reachable: a TAP-Vid-shaped pickle of a seeded synthetic
256 x 256 sequence stands in for it (mft_amd.tapvid.synthetic_pickle)."""
import pickle
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import mft_oracle as O
from mft_amd import tapvid
from mft_amd.config import Config, load_config
from mft_amd.io import FlowCache
from mft_amd.results import FlowOUTrackingResult
from mft_amd.synth import SyntheticVideo
from mft_amd.weights import make_weights

pytestmark = pytest.mark.gpu
REPO = __import__("pathlib").Path(__file__).resolve().parents[1]
DELTAS = [np.inf, 1, 2, 4]
ITERS = 12


class CountingCache(FlowCache):
    """The product's cache (HBM tier) with its traffic counted."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.reads, self.hits, self.writes = [], 0, []

    def read(self, left_id, right_id):
        val = super().read(left_id, right_id)
        self.reads.append((left_id, right_id))
        self.hits += val[0] is not None
        return val

    def write(self, left_id, right_id, *val):
        self.writes.append((left_id, right_id))
        return super().write(left_id, right_id, *val)


class OracleCache:
    """What the reference's FlowCache does for the protocol (exact entries, MFT/utils/io.py:655-700), counted the same way."""

    def __init__(self):
        self.store, self.reads, self.hits, self.writes = {}, [], 0, []

    def read(self, l, r):
        self.reads.append((l, r))
        if (l, r) in self.store:
            self.hits += 1
            return self.store[(l, r)]
        return None, None, None

    def write(self, l, r, *val):
        self.writes.append((l, r))
        self.store[(l, r)] = val

    def clear(self):
        self.store.clear()


class OracleMFT:
    """The oracle tracker (oracle/mft_oracle.py: MFT.init / track restated) behind the tracker API the runner drives, with the
    reference's cache protocol around its flow calls (MFT/MFT.py:99-102, 205-231: finite deltas read and write the cache, the
    delta = infinity candidate -- the one whose left frame is the start frame -- bypasses it)."""

    def __init__(self, sd, log, deltas=None, iters=ITERS):
        self.sd, self.log = sd, log
        self.deltas, self.iters = list(deltas if deltas is not None else DELTAS), iters
        self.memo = {}                      # (hash of the video is implicit: one instance per sequence)

    def _raft(self, l, r, li, ri):
        if (l, r) not in self.memo:
            with torch.no_grad():
                self.memo[(l, r)] = O.compute_flow(self.sd, li, ri, self.iters)
        return self.memo[(l, r)]

    def _flow(self, l, r, li, ri):
        use_cache = l != self.tr.start
        self.frame_pairs.append((l, r))
        if use_cache:
            got = self.cache.read(l, r)
            if got[0] is not None:
                return got
        val = self._raft(l, r, li, ri)
        if use_cache:
            self.cache.write(l, r, *val)
        return val

    def init(self, img, start_frame_i=0, time_direction=1, flow_cache=None, **kw):
        self.cache = flow_cache
        self.tr = O.Tracker(self._flow, deltas=self.deltas, occlusion_threshold=0.02)
        m = self.tr.init(img, start_frame_i, time_direction)
        return SimpleNamespace(result=FlowOUTrackingResult(*m.result))

    def track(self, img, debug=False, **kw):
        self.frame_pairs = []
        with torch.no_grad():
            m = self.tr.track(img)
        self.log.append(sorted(self.frame_pairs))
        return SimpleNamespace(result=FlowOUTrackingResult(*m.result))


def test_c3_tapvid_protocol_vs_oracle(tmp_path):
    torch.set_num_threads(16)
    # ---- the dataset: one 10-frame 256 x 256 sequence, 14 tracks, as a TAP-Vid pickle
    vid = SyntheticVideo(256, 256, n_frames=10, seed=77)
    tapvid.synthetic_pickle(tmp_path / "synthetic_davis.pkl", {"synth-a": vid}, n_tracks=14, seed=3)
    dconf = Config()
    dconf.pickles, dconf.scaling, dconf.name = [tmp_path / "synthetic_davis.pkl"], "256x256_512x512", "synthetic-256x256_512x512"
    (el,) = list(tapvid.create_tapvid_dataset(dconf.pickles[0], ["first", "strided"], dconf.scaling))
    assert el["data"]["first"]["video"].shape == (1, 10, 512, 512, 3)             # tracked at 512 x 512
    # ---- HIP tracker through the dataset runner (what tools/run_MFT_tapvid.py does)
    conf = load_config(REPO / "configs" / "MFT_cfg.py")
    conf.flow_config.model = None
    conf.flow_config.synthetic_weights_seed = 0
    conf.flow_config.flow_iters = ITERS
    conf.deltas = list(DELTAS)
    conf.keep_result_on_device = True
    tracker = conf.tracker_class(conf)
    hip_pairs = []
    orig_track = tracker.track

    def track(img, **kw):
        m = orig_track(img, **kw)
        hip_pairs.append(sorted(tracker.last_pairs))
        return m
    tracker.track = track
    caches = []

    def factory(d, ram, gpu):
        caches.append(CountingCache(d, max_RAM_MB=ram, max_GPU_RAM_MB=gpu))
        return caches[-1]
    done = tapvid.run_dataset(dconf, [conf], tmp_path / "export", tmp_path / "cache", mode="both", tracker=tracker,
                              gpu_cache_limit=64, cache_factory=factory)
    assert [(d["mode"], d["skipped"]) for d in done] == [("first", False), ("strided", False)] and len(caches) == 1
    hip = {}
    for d in done:
        with open(d["path"], "rb") as f:
            hip[d["mode"]] = pickle.load(f)
    # ---- the same protocol on the oracle tracker (CPU), same video, same queries, one cache
    sd = {k: torch.from_numpy(v) for k, v in make_weights(0).items()}
    ora_pairs = []
    ora = OracleMFT(sd, ora_pairs)
    ocache = OracleCache()
    video = np.ascontiguousarray(el["data"]["first"]["video"][0][..., ::-1])
    want = {}
    for mode in ("first", "strided"):                                              # (the runner's order)
        q = np.asarray(el["data"][mode]["query_points"])[0].astype(np.int64)
        want[mode] = tapvid.run_sequence(ora, video, q, mode, flow_cache=ocache)
    # ---- identical requests: per tracked frame the same (left, right) pairs, the same cache reads / hits / writes
    assert hip_pairs == ora_pairs and len(hip_pairs) >= 20
    assert caches[0].reads == ocache.reads and caches[0].writes == ocache.writes and caches[0].hits == ocache.hits
    assert ocache.hits >= 5                                                         # the protocol does re-request pairs
    # ---- tracks and occlusion scores on the 256 x 256 raster
    tol = 1e-3 * 256 / 512
    for mode in ("first", "strided"):
        assert hip[mode]["tracks"].shape == want[mode]["tracks"].shape == el["data"][mode]["target_points"].shape
        d = np.abs(hip[mode]["tracks"] - want[mode]["tracks"]).max(-1)
        assert (d <= tol).mean() >= 0.99, (mode, float((d <= tol).mean()), float(d.max()))
        do = np.abs(hip[mode]["occluded"] - want[mode]["occluded"])
        assert (do <= 1e-4).mean() >= 0.99, (mode, float((do <= 1e-4).mean()), float(do.max()))
    # ---- and through the evaluation: the HIP run and the oracle run score the same
    m_hip = tapvid.evaluate_dataset(dconf, [conf], tmp_path / "export", mode="both", write=True)
    for mode in ("first", "strided"):
        with open(tapvid.result_path(tmp_path / "export", conf.name, "synth-a", mode), "wb") as f:
            pickle.dump(want[mode], f)
    m_ora = tapvid.evaluate_dataset(dconf, [conf], tmp_path / "export", mode="both", write=False)
    for mode in ("first", "strided"):
        a, b = m_hip[mode][conf.name][0], m_ora[mode][conf.name][0]
        for k in ("average_jaccard", "average_pts_within_thresh", "occlusion_accuracy"):
            assert abs(a[k] - b[k]) <= 0.01, (mode, k, a[k], b[k])
    assert (tmp_path / "export" / conf.name / "eval" / "tapvid-eval-strided.pklz").exists()


@pytest.mark.timeout(900)
def test_c3_all_seven_deltas_strided_forward_and_backward_vs_oracle(tmp_path):
    """The protocol with the reference's FULL delta set: a 128 x 128 sequence of 40 frames through the
    reader's '256x256' scaling (tracked on the 256 raster), 'strided' queries, i.e. every start frame tracked forward AND
    backward, all of {inf, 1, 2, 4, 8, 16, 32} live (delta 32 from the 33rd frame of a run on: forward from frame 0, backward
    from frame 35), one flow cache shared by all runs -- the HIP tracker against the oracle tracker: the same (left, right) pairs
    per tracked frame, the same cache reads / hits / writes, tracks and occlusion scores on the 256 raster.  The query frames
    are chosen (0 and 35) instead of sampled every fifth frame: 74 tracked frames and ~430 distinct oracle pairs instead of
    8 x 39 frames."""
    torch.set_num_threads(16)
    deltas = [np.inf, 1, 2, 4, 8, 16, 32]
    vid = SyntheticVideo(128, 128, n_frames=40, seed=41)
    tapvid.synthetic_pickle(tmp_path / "d.pkl", {"seq": vid}, n_tracks=10, seed=9)
    (el,) = list(tapvid.create_tapvid_dataset(tmp_path / "d.pkl", ["strided"], "256x256"))
    video = np.ascontiguousarray(el["data"]["strided"]["video"][0][..., ::-1])
    assert video.shape == (40, 256, 256, 3)
    rng = np.random.default_rng(3)
    queries = np.concatenate([np.stack([np.full(6, t), rng.integers(8, 248, 6), rng.integers(8, 248, 6)], 1) for t in (0, 35)])
    conf = load_config(REPO / "configs" / "MFT_cfg.py")
    conf.flow_config.model, conf.flow_config.synthetic_weights_seed, conf.flow_config.flow_iters = None, 0, ITERS
    assert list(conf.deltas) == deltas                                            # the shipped configuration IS the full set
    conf.keep_result_on_device = True
    tracker = conf.tracker_class(conf)
    hip_pairs = []
    orig_track = tracker.track

    def track(img, **kw):
        m = orig_track(img, **kw)
        hip_pairs.append(sorted(tracker.last_pairs))
        return m
    tracker.track = track
    cache = CountingCache(tmp_path / "cache", max_RAM_MB=512, max_GPU_RAM_MB=1024)
    hip = tapvid.run_sequence(tracker, video, queries, "strided", flow_cache=cache, device="cuda")
    sd = {k: torch.from_numpy(v) for k, v in make_weights(0).items()}
    ora_pairs = []
    ocache = OracleCache()
    want = tapvid.run_sequence(OracleMFT(sd, ora_pairs, deltas=deltas), video, queries, "strided", flow_cache=ocache)
    assert hip_pairs == ora_pairs and len(hip_pairs) == 39 + 4 + 35
    assert max(len(p) for p in hip_pairs) == 7                                      # every delta live, in both directions
    assert sum(len(p) == 7 for p in hip_pairs[:39]) == 7 and sum(len(p) == 7 for p in hip_pairs[43:]) == 3
    assert cache.reads == ocache.reads and cache.writes == ocache.writes and cache.hits == ocache.hits
    assert ocache.hits >= 3
    d = np.abs(hip["tracks"] - want["tracks"]).max(-1)
    assert (d <= 1e-3).mean() >= 0.99, (float((d <= 1e-3).mean()), float(d.max()))
    do = np.abs(hip["occluded"] - want["occluded"])
    assert (do <= 1e-4).mean() >= 0.99, (float((do <= 1e-4).mean()), float(do.max()))


def test_c3_runner_cont_and_flow_export(tmp_path):
    """The runner's file protocol on the device tracker: `cont` recomputes nothing, `write_flow` exports the frame-0 template's
    results in the reference's X16 cache-entry format (run_MFT_tapvid.py:160-165, 214-221)."""
    vid = SyntheticVideo(128, 160, n_frames=6, seed=5)
    data = tapvid.synthetic_pickle(tmp_path / "p.pkl", {"s": vid}, n_tracks=6, seed=1)
    data["s"]["occluded"][0, 0] = False                                          # a track visible in frame 0: start frame 0 exists
    with open(tmp_path / "p.pkl", "wb") as f:
        pickle.dump(data, f)
    dconf = Config()
    dconf.pickles, dconf.scaling, dconf.name = [tmp_path / "p.pkl"], "fullres", "p"
    conf = load_config(REPO / "configs" / "MFT_cfg.py")
    conf.flow_config.model, conf.flow_config.synthetic_weights_seed, conf.flow_config.flow_iters = None, 0, 4
    conf.deltas = [np.inf, 1, 2]
    tracker = conf.tracker_class(conf)
    done = tapvid.run_dataset(dconf, [conf], tmp_path / "e", tmp_path / "c", mode="first", tracker=tracker, write_flow=True)
    assert len(done) == 1 and not done[0]["skipped"]
    files = sorted((tmp_path / "e" / conf.name / "flowous" / "s").glob("0--*.flowouX16.pkl"))
    assert len(files) == 6
    r = FlowOUTrackingResult.read(files[2])
    assert r.flow.shape == (2, 128, 160) and torch.isfinite(r.flow).all()
    n = len(tracker.flower._frames)
    again = tapvid.run_dataset(dconf, [conf], tmp_path / "e", tmp_path / "c", mode="first", cont=True, tracker=tracker)
    assert again[0]["skipped"] and len(tracker.flower._frames) == n


def test_run_MFT_tapvid_command_line(tmp_path):
    """tools/run_MFT_tapvid.py with the reference's arguments on a synthetic TAP-Vid-shaped pickle: dataset config with the
    '256x256_512x512' scaling -> tracklet pickles -> evaluation, one JSON summary."""
    import json
    import subprocess
    import sys
    out = subprocess.run([sys.executable, str(REPO / "tools" / "run_MFT_tapvid.py"), str(REPO / "dataset_configs" / "pkl-tapvid-davis-256x256_512x512.py"),
                          str(REPO / "configs" / "MFT_cfg.py"), "--synthetic", "1", "--synthetic-frames", "8", "--export", str(tmp_path / "e"),
                          "--cache", str(tmp_path / "c"), "--mode", "both"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["scaling"] == "256x256_512x512" and d["results"] == 2 and d["skipped"] == 0
    assert sorted(d["metrics"]) == ["first", "strided"]
    m = d["metrics"]["strided"]["MFT_cfg"]
    assert 0.0 <= m["average_jaccard"] <= 1.0 and 0.0 <= m["occlusion_accuracy"] <= 1.0
    assert (tmp_path / "e" / "MFT_cfg" / "results" / "synth-00-first.pklz").exists()
    assert (tmp_path / "e" / "MFT_cfg" / "eval" / "tapvid-eval.pklz").exists()
