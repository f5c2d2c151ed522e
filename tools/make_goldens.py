#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (build container only).

    python tools/make_goldens.py            # needs /root/reference (read-only)

The reference is imported from ``/root/reference`` with the shims of SURVEY.md
section 8(c); nothing of it is copied: only its numeric OUTPUTS on seeded inputs
(``tests/golden_inputs.py``) are stored.  The GPU box never sees the reference.

Shims (none touch /root/reference):
  * ``cv2`` is absent -> stub module (imported at top level by MFT.utils.io /
    geom_utils, never called on the hot path);
  * no GPU here and 'cuda' is hard-coded (MFT/MFT.py:20, MFT/raft.py:17,45) ->
    ``Tensor.cuda`` / ``.to('cuda')`` become no-ops, ``identity(device='cuda')``
    is redirected to cpu;
  * the trained checkpoint is missing -> ``RAFT.load_state_dict`` from the
    build's seeded generator (``mft_amd.weights.make_weights``).
"""
import sys
import types
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
REF = Path("/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
sys.path.insert(1, str(REF))

cv2 = types.ModuleType("cv2")
cv2.INTER_NEAREST = 0
cv2.IMREAD_UNCHANGED = -1
# The codec pin (gen_codec) needs to see what the reference hands to / takes from cv2's PNG coder: the stub
# keeps the array given to imencode and returns it from imdecode (PNG is lossless, so this IS cv2's round trip).
CV2_CAPTURED = []


def _imencode(ext, arr):
    assert ext == ".png"
    CV2_CAPTURED.append(np.array(arr, copy=True))
    return True, np.array([len(CV2_CAPTURED) - 1], np.int64)


def _imdecode(buf, flags):
    return CV2_CAPTURED[int(np.asarray(buf).reshape(-1)[0])].copy()


CV2_FILES = {}


def _imwrite(path, arr):
    CV2_FILES[str(path)] = np.array(arr, copy=True)
    return True


def _imread(path, flags=None):
    return CV2_FILES[str(path)].copy()


cv2.IMREAD_ANYDEPTH = 2
cv2.imencode, cv2.imdecode, cv2.imwrite, cv2.imread = _imencode, _imdecode, _imwrite, _imread
sys.modules["cv2"] = cv2

import torch  # noqa: E402

torch.set_num_threads(8)
_orig_to = torch.Tensor.to


def _to(self, *a, **k):
    a = tuple("cpu" if (isinstance(x, str) and x.startswith("cuda")) else x for x in a)
    if isinstance(k.get("device"), str) and k["device"].startswith("cuda"):
        k["device"] = "cpu"
    return _orig_to(self, *a, **k)


torch.Tensor.to = _to
torch.Tensor.cuda = lambda self, *a, **k: self
_orig_zeros = torch.zeros


def _zeros(*a, **k):
    if isinstance(k.get("device"), str) and k["device"].startswith("cuda"):
        k["device"] = "cpu"
    return _orig_zeros(*a, **k)


torch.zeros = _zeros

from MFT.MFT import MFT, chain_results  # noqa: E402
from MFT.raft import RAFTWrapper  # noqa: E402
from MFT.results import FlowOUTrackingResult  # noqa: E402
from MFT.config import Config  # noqa: E402
from MFT.RAFT.core.raft import RAFT  # noqa: E402
from MFT.RAFT.core.corr import CorrBlock  # noqa: E402

import golden_inputs as gi  # noqa: E402
from mft_amd.weights import make_weights  # noqa: E402
from mft_amd.synth import SyntheticVideo  # noqa: E402

OUT = REPO / "tests" / "golden"
OUT.mkdir(parents=True, exist_ok=True)


class AttrDict(dict):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.__dict__.update(k)


def build_reference_model():
    args = AttrDict(occlusion_module="separate_with_uncertainty", small=False, mixed_precision=False)
    model = RAFT(args)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in make_weights(gi.WEIGHT_SEED).items()}
    missing = model.load_state_dict(sd, strict=True)
    print("load_state_dict:", missing)
    model.requires_grad_(False)
    model.eval()
    return model


def build_reference_flower(model, iters):
    fl = object.__new__(RAFTWrapper)
    fl.C = Config()
    fl.C.flow_iters = iters
    fl.model = model
    return fl


def build_reference_tracker(flower, deltas=(np.inf, 1, 2, 4, 8, 16, 32), thr=0.02):
    tr = object.__new__(MFT)
    tr.C = Config()
    tr.C.deltas = list(deltas)
    tr.C.occlusion_threshold = thr
    tr.flower = flower
    tr.device = "cpu"
    return tr


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def N(x):
    return x.detach().cpu().numpy().astype(np.float32)


def gen_ops(model):
    d = {k: T(v) for k, v in gi.ops_inputs().items()}
    out = {}
    with torch.no_grad():
        cb = CorrBlock(d["fmap1"], d["fmap2"], radius=4)
        for l, lvl in enumerate(cb.corr_pyramid):
            out[f"pyr{l}_checksum"] = gi.checksum(N(lvl))
            out[f"pyr{l}_shape"] = np.array(lvl.shape)
        rows = [0, 5, 100, 383]
        for l, lvl in enumerate(cb.corr_pyramid):
            out[f"pyr{l}_rows"] = N(lvl[rows])
        out["pyr_rows_idx"] = np.array(rows)
        out["lookup"] = N(cb(d["coords1"]))
        ub = model.update_block
        net, mask, delta, motion = ub(d["net"], d["inp"], d["corr"], d["flow"])
        out["ub_net"], out["ub_mask"], out["ub_delta"], out["ub_motion"] = map(N, (net, mask, delta, motion))
        occl, unc = model.occlusion_block(d["net"], d["inp"], d["corr"], d["flow"], d["delta_flow"], d["motion"])
        out["ou_occl"], out["ou_unc"] = N(occl), N(unc)
        out["up_flow"] = N(model.upsample_flow(d["flow"], d["mask"]))
        out["up_occl"] = N(model.upsample_flow(d["occl_lr"], d["mask"], mult_coef=1.0))
        out["up_unc"] = N(model.upsample_flow(d["unc_lr"], d["mask"], mult_coef=1.0, n_channels=1))
        # encoders on a small image pair
        vid = SyntheticVideo(128, 192, n_frames=4, seed=3)
        img = T(vid[0][:, :, ::-1].copy()).permute(2, 0, 1)[None].float()
        x = 2 * (img / 255.0) - 1.0
        out["fnet"] = N(model.fnet(x))
        out["cnet"] = N(model.cnet(x))
    np.savez_compressed(OUT / "ops.npz", **out)
    print("ops.npz", sum(v.nbytes for v in out.values()) / 1e6, "MB")


def gen_compute_flow(model):
    out = {}
    with torch.no_grad():
        for tag, (H, W, iters, fa, fb) in {"a": (128, 192, 12, 0, 3), "b": (125, 187, 4, 1, 2)}.items():
            vid = SyntheticVideo(H, W, n_frames=8, seed=5)
            fl = build_reference_flower(model, iters)
            flow, extra = fl.compute_flow(vid[fa], vid[fb], mode="flow")
            out[f"{tag}_flow"], out[f"{tag}_occl"], out[f"{tag}_sigma"] = N(flow), N(extra["occlusion"]), N(extra["sigma"])
            out[f"{tag}_meta"] = np.array([H, W, iters, fa, fb])
            print(tag, "flow mean abs", float(flow.abs().mean()), "occl>thr", float((extra["occlusion"] > 0.02).float().mean()),
                  "sigma mean", float(extra["sigma"].mean()))
        # (c) with an initial flow (MFT/raft.py:49-52 -> core/raft.py:153-154)
        H, W, iters, fa, fb = 125, 187, 3, 0, 2
        vid = SyntheticVideo(H, W, n_frames=8, seed=5)
        fl = build_reference_flower(model, iters)
        flow, extra = fl.compute_flow(vid[fa], vid[fb], mode="flow", init_flow=T(gi.init_flow_input(H, W)))
        out["c_flow"], out["c_occl"], out["c_sigma"] = N(flow), N(extra["occlusion"]), N(extra["sigma"])
        out["c_meta"] = np.array([H, W, iters, fa, fb])
    np.savez_compressed(OUT / "compute_flow.npz", **out)


class StubFlower:
    """compute_flow() that returns tests/golden_inputs.stub_flowou for the pair
    encoded in the two images."""

    def compute_flow(self, src_img, dst_img, mode="flow", init_flow=None, **kw):
        l, r = gi.decode_id(src_img), gi.decode_id(dst_img)
        flow, occl, sigma = gi.stub_flowou(l, r)
        return T(flow), {"occlusion": T(occl), "sigma": T(sigma), "debug": None}


class TrackRecorder:
    """Records, for one reference MFT.track() call, the (left_id, right_id) pairs it requests
    (get_flowou_with_cache, MFT/MFT.py:99-102) and the chained candidates in delta order (chain_results,
    MFT/MFT.py:104), and recomputes the per-pixel selected candidate exactly like MFT/MFT.py:112-124
    (stack -> -sigma -> -inf where occluded -> max(dim=0).indices).  The reference's functions are wrapped,
    not changed."""

    def __init__(self):
        import MFT.MFT as refmod
        self.refmod = refmod
        self.orig_chain, self.orig_get = refmod.chain_results, refmod.get_flowou_with_cache
        self.pairs, self.cands = [], []

        def chain(L, R):
            c = self.orig_chain(L, R)
            self.cands.append(c)
            return c

        def get(flower, left_img, right_img, flow_init=None, cache=None, left_id=None, right_id=None, **kw):
            self.pairs.append((int(left_id), int(right_id)))
            return self.orig_get(flower, left_img, right_img, flow_init, cache, left_id, right_id, **kw)

        refmod.chain_results, refmod.get_flowou_with_cache = chain, get

    def close(self):
        self.refmod.chain_results, self.refmod.get_flowou_with_cache = self.orig_chain, self.orig_get

    def begin(self):
        self.pairs, self.cands = [], []

    def chosen(self, deltas, thr):
        """deltas: the deltas that produced self.cands, in call order."""
        order = sorted(range(len(deltas)), key=lambda i: 0 if np.isinf(deltas[i]) else deltas[i])
        res = [self.cands[i] for i in order]
        sig = torch.stack([r.sigma for r in res], 0)
        occ = torch.stack([r.occlusion for r in res], 0)
        scores = -sig
        scores[occ > thr] = -float("inf")
        return scores.max(dim=0, keepdim=True).indices[0, 0].numpy().astype(np.int8)


def live_deltas(tr, frame_i):
    """The deltas whose candidate MFT.track builds at frame_i, in C.deltas order (MFT/MFT.py:74-91)."""
    out, used = [], []
    for d in tr.C.deltas:
        left = frame_i - d * tr.time_direction
        if tr.is_before_start(left):
            if not np.isinf(d):
                continue
            left = tr.start_frame_i
        if int(left) in used:
            continue
        used.append(int(left))
        out.append(d)
    return out


def gen_chain_and_sequence():
    out = {}
    # (1) bare chain_results on two stub results
    L = FlowOUTrackingResult(*map(T, gi.stub_flowou(0, 7)))
    R = FlowOUTrackingResult(*map(T, gi.stub_flowou(7, 9)))
    c = chain_results(L, R)
    out["chain_flow"], out["chain_occl"], out["chain_sigma"] = N(c.flow), N(c.occlusion), N(c.sigma)
    out["chain_invalid"] = c.invalid_mask().numpy()
    # (2) full init/track sequences with the stub flower: forward and backward
    keep = [1, 2, 3, 5, 9, 17, 33, 34, 43]
    rec = TrackRecorder()
    for tag, (start, direction) in {"fwd": (0, +1), "bwd": (gi.SEQ_FRAMES - 1, -1)}.items():
        tr = build_reference_tracker(StubFlower())
        tr.init(gi.id_image(start), start_frame_i=start, time_direction=direction)
        sums, keys, pairs = [], [], []
        for step in range(1, gi.SEQ_FRAMES):
            fid = start + direction * step
            rec.begin()
            meta = tr.track(gi.id_image(fid))
            res = meta.result
            pairs.append(np.array([l for l, r in rec.pairs] + [-1] * (7 - len(rec.pairs))))
            assert all(r == fid for l, r in rec.pairs)
            if step in keep:
                out[f"{tag}_{step}_chosen"] = rec.chosen(live_deltas(tr, fid), tr.C.occlusion_threshold)
            sums.append(np.concatenate([gi.checksum(N(res.flow)), gi.checksum(N(res.occlusion)), gi.checksum(N(res.sigma))]))
            keys.append(np.array(sorted(tr.memory.keys()) + [-1] * (40 - len(tr.memory))))
            if step in keep:
                out[f"{tag}_{step}_flow"], out[f"{tag}_{step}_occl"], out[f"{tag}_{step}_sigma"] = \
                    N(res.flow), N(res.occlusion), N(res.sigma)
        out[f"{tag}_checksums"] = np.stack(sums)
        out[f"{tag}_memory_keys"] = np.stack(keys)
        out[f"{tag}_keep"] = np.array(keep)
        out[f"{tag}_left_ids"] = np.stack(pairs)          # requested left ids per step, in request order, -1 padded
    rec.close()
    np.savez_compressed(OUT / "sequence_stub.npz", **out)
    print("sequence_stub.npz", sum(v.nbytes for v in out.values()) / 1e6, "MB")


def gen_e2e(model):
    out = {}
    vid = SyntheticVideo(gi.E2E_H, gi.E2E_W, n_frames=gi.E2E_FRAMES, seed=11)
    fl = build_reference_flower(model, gi.E2E_ITERS)
    requested = []
    orig = fl.compute_flow

    tr = build_reference_tracker(fl)
    keep = [1, 3, 9, 33, 41]
    rec = TrackRecorder()
    with torch.no_grad():
        tr.init(vid[0])
        sums, pairs = [], []
        for i in range(1, gi.E2E_FRAMES):
            rec.begin()
            meta = tr.track(vid[i])
            res = meta.result
            pairs.append(np.array([l for l, r in rec.pairs] + [-1] * (7 - len(rec.pairs))))
            if i in keep:
                out[f"f{i}_chosen"] = rec.chosen(live_deltas(tr, i), tr.C.occlusion_threshold)
            sums.append(np.concatenate([gi.checksum(N(res.flow)), gi.checksum(N(res.occlusion)), gi.checksum(N(res.sigma))]))
            if i in keep:
                out[f"f{i}_flow"], out[f"f{i}_occl"], out[f"f{i}_sigma"] = N(res.flow), N(res.occlusion), N(res.sigma)
            if i % 10 == 0:
                print("e2e frame", i, "occl frac", float((res.occlusion > 0.5).float().mean()),
                      "flow mean abs", float(res.flow.abs().mean()))
    rec.close()
    out["checksums"] = np.stack(sums)
    out["keep"] = np.array(keep)
    out["left_ids"] = np.stack(pairs)
    np.savez_compressed(OUT / "sequence_raft.npz", **out)
    print("sequence_raft.npz", sum(v.nbytes for v in out.values()) / 1e6, "MB")


def gen_results_api():
    """FlowOUTrackingResult's point-query and warping methods (MFT/results.py:116-265) on seeded inputs."""
    d = gi.results_api_inputs()
    res = FlowOUTrackingResult(T(d["flow"]), T(d["occl"]), T(d["sigma"]))
    out = {}
    out["warp_forward"] = res.warp_forward(torch.from_numpy(d["img"]))
    out["warp_forward_masked"] = res.warp_forward(torch.from_numpy(d["img"]), mask=d["mask"], border=-1.0)
    out["warp_forward_points"] = N(res.warp_forward_points(T(d["pts"])))
    f, o, s_ = res.sample(T(d["pts"]))
    out["sample_flow"], out["sample_occl"], out["sample_sigma"] = N(f), N(o), N(s_)
    out["warp_backward"] = N(res.warp_backward(torch.from_numpy(d["img"]).permute(2, 0, 1).contiguous()))
    out["invalid_mask"] = res.invalid_mask().numpy()
    np.savez_compressed(OUT / "results_api.npz", **out)
    print("results_api.npz", sum(v.nbytes for v in out.values()) / 1e3, "kB")


def gen_codec():
    """The flow-cache codec of the reference itself (write_flowou_X16 / read_flowou_X16, MFT/utils/io.py:495-563)
    on seeded inputs: the uint8 (B, G, R) planes it hands to cv2.imencode, the per-channel (min, max) it pickles,
    and what read_flowou_X16 makes of them again."""
    import tempfile
    from MFT.utils import io as refio
    d = gi.codec_inputs()
    out = {}
    del CV2_CAPTURED[:]
    with tempfile.TemporaryDirectory() as tmp:
        path = str(Path(tmp) / "3--5.flowouX16.pkl")
        refio.write_flowou_X16(path, d["flow"], d["occl"], d["sigma"])
        import pickle
        with open(path, "rb") as f:
            pk = pickle.load(f)
        names = ("flow_x", "flow_y", "occlusion", "sigma")
        assert len(CV2_CAPTURED) == 4
        out["bgr"] = np.stack(CV2_CAPTURED)                                   # [4, H, W, 3] uint8
        out["lohi"] = np.array([[pk[n]["min"], pk[n]["max"]] for n in names], np.float32)
        flow, occl, sigma = refio.read_flowou_X16(path)
        out["dec_flow"], out["dec_occl"], out["dec_sigma"] = (np.asarray(a, np.float32) for a in (flow, occl, sigma))
        assert flow.dtype == np.float32
        # the other two formats (MFT/utils/io.py:222-291, 372-443)
        p1 = Path(tmp) / "3--5.flowou.png"
        refio.write_flowou1_png(p1, d["flow"], d["occl"], d["sigma"])
        out["png16_bgra"] = CV2_FILES[str(p1)]                                 # uint16 [H, W, 4] handed to cv2.imwrite
        f1, o1, s1 = refio.read_flowou1_png(p1)
        out["png16_dec_flow"], out["png16_dec_occl"], out["png16_dec_sigma"] = (np.asarray(a) for a in (f1, o1, s1))
        del CV2_CAPTURED[:]
        p2 = str(Path(tmp) / "3--5.flowouX32.pkl")
        refio.write_flowou_X32(p2, d["flow"], d["occl"], d["sigma"])
        out["x32_bgra"] = np.stack(CV2_CAPTURED)                               # [4, H, W, 4] uint8
        with open(p2, "rb") as f:
            pk = pickle.load(f)
        out["x32_lohi"] = np.array([[pk[n]["min"], pk[n]["max"]] for n in names], np.float32)
        f2, o2, s2 = refio.read_flowou_X32(p2)
        out["x32_dec_flow"], out["x32_dec_occl"], out["x32_dec_sigma"] = (np.asarray(a) for a in (f2, o2, s2))
    np.savez_compressed(OUT / "codec.npz", **out)
    print("codec.npz", sum(v.nbytes for v in out.values()) / 1e3, "kB")


def gen_tapvid():
    """TAP-Vid query samplers and metrics of the reference on seeded random tracks."""
    for name in ("mediapy", "PIL", "PIL.Image"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["PIL"].Image = sys.modules["PIL.Image"]
    from MFT.evaluation import tapvid_eval_stuff as tves
    d = gi.tapvid_inputs()
    out = {}
    for mode in ("first", "strided"):
        for v in range(2):
            # per video: sample queries the reference's way, then score the predictions of the sampled tracks
            occ, pts = d["gt_occluded"][v], d["gt_tracks"][v]
            s = (tves.sample_queries_first(occ, pts, d["frames"]) if mode == "first"
                 else tves.sample_queries_strided(occ, pts, d["frames"], query_stride=5))
            for k in ("query_points", "target_points", "occluded", "trackgroup"):
                out[f"{mode}_{v}_{k}"] = np.asarray(s[k])
            tg = s["trackgroup"][0]
            m = tves.compute_tapvid_metrics(s["query_points"], s["occluded"], s["target_points"],
                                            d["pred_occluded"][v][tg][None], d["pred_tracks"][v][tg][None], mode)
            for k, val in m.items():
                out[f"{mode}_{v}_metric_{k}"] = np.asarray(val)
        # and the batched call (both videos at once, all tracks, queries on frame 0)
        q = np.zeros((2, d["gt_tracks"].shape[1], 3))
        occ = d["gt_occluded"].copy()
        occ[:, :, 0] = False
        m = tves.compute_tapvid_metrics(q, occ, d["gt_tracks"], d["pred_occluded"], d["pred_tracks"], mode)
        for k, val in m.items():
            out[f"{mode}_batch_metric_{k}"] = np.asarray(val)
    np.savez_compressed(OUT / "tapvid.npz", **out)
    print("tapvid.npz", sum(v.nbytes for v in out.values()) / 1e3, "kB,", len(out), "arrays")


def gen_tapvid_dataset():
    """The reference's dataset reader (create_tapvid_dataset: tapvid_eval_stuff.py:612-672, parse_scale_WH: MFT/utils/misc.py:65-92,
    load_kinetics_video: :528-549) on two small TAP-Vid-shaped pickles written here as FIXTURES (tests/golden/tapvid_*.pkl:
    data -- a dict-form pickle and a Kinetics-style list of JPEG-encoded sequences) under every kind of scaling string.
    ``mediapy`` is absent: its ``resize_video`` is stubbed with what mediapy does for uint8 RGB frames -- PIL's Lanczos filter,
    frame by frame -- so the resampler itself is pinned only as far as that stub is faithful; everything around it (the chain of
    rescalings, which size the points are scaled to, the train_size carry-over between sequences, the shard naming, the in-place
    point scaling, the samplers' inputs) is the reference's own code running."""
    import io as pyio
    import pickle
    import zlib
    from PIL import Image
    mediapy = types.ModuleType("mediapy")

    def resize_video(video, shape):
        return np.stack([np.asarray(Image.fromarray(f).resize((shape[1], shape[0]), resample=Image.Resampling.LANCZOS))
                         for f in video])
    mediapy.resize_video = resize_video
    sys.modules["mediapy"] = mediapy
    from MFT.evaluation import tapvid_eval_stuff as tves
    from MFT.utils.misc import parse_scale_WH
    seqs = gi.tapvid_pickle_sequences()
    dict_path = OUT / "tapvid_davis_like.pkl"
    with open(dict_path, "wb") as f:
        pickle.dump(seqs, f, protocol=4)
    kin = []
    for name, d in seqs.items():
        jpegs = []
        for frame in d["video"]:
            buf = pyio.BytesIO()
            Image.fromarray(frame).save(buf, format="JPEG", quality=92)
            jpegs.append(buf.getvalue())
        kin.append({"video": jpegs, "points": d["points"].copy(), "occluded": d["occluded"].copy()})
    kin_path = OUT / "tapvid_kinetics_like_0007.pkl"
    with open(kin_path, "wb") as f:
        pickle.dump(kin, f, protocol=4)
    out = {}
    for form, path in (("dict", dict_path), ("kin", kin_path)):
        for sc in gi.TAPVID_SCALINGS:
            arg = {"default": None, "false": False}.get(sc, sc)
            for fake in (False, True):
                if fake and sc not in ("256x256_512x512", "x30"):
                    continue
                for i, el in enumerate(tves.create_tapvid_dataset(path, ["first", "strided"], arg, fake_video=fake)):
                    key = f"{form}|{sc}|{'fake' if fake else 'real'}|{i}"
                    out[key + "|name"] = np.array(el["video_name"])
                    out[key + "|N"] = np.array(el["N_sequences"])
                    for mode in ("first", "strided"):
                        g = el["data"][mode]
                        video = np.asarray(g["video"])
                        out[f"{key}|{mode}|video_shape"] = np.array(video.shape)
                        out[f"{key}|{mode}|video_crc"] = np.array(zlib.crc32(np.ascontiguousarray(video).tobytes()))
                        if mode == "first" and video.size <= 60000:
                            out[f"{key}|video"] = video
                        for k in ("query_points", "target_points", "occluded", "trackgroup"):
                            out[f"{key}|{mode}|{k}"] = np.asarray(g[k])
    shapes = [{"N_frames": 8, "H": 40, "W": 56, "C": 3}, {"N_frames": 3, "H": 480, "W": 854, "C": 3}]
    for j, fs in enumerate(shapes):
        for sc in ("fullres", "256x256", "x1080", "512x", "256x256_x480", "fullres_300x", "x7_9x_fullres"):
            res = parse_scale_WH(sc, fs)
            out[f"parse|{j}|{sc}"] = np.array([[d["N_frames"], d["H"], d["W"], d["C"]] for d in res])
    np.savez_compressed(OUT / "tapvid_dataset.npz", **out)
    print("tapvid_dataset.npz", sum(v.nbytes for v in out.values()) / 1e3, "kB,", len(out), "arrays;",
          dict_path.stat().st_size / 1e3, "+", kin_path.stat().st_size / 1e3, "kB of fixture pickles")


if __name__ == "__main__":
    assert REF.exists(), "the reference is only mounted in the build container"
    if sys.argv[1:] == ["tapvid_dataset"]:
        gen_tapvid_dataset()
        sys.exit(0)
    which = sys.argv[1:] or ["ops", "flow", "seq", "e2e", "tapvid", "results", "codec"]
    if set(which) <= {"tapvid", "results", "codec"}:
        if "codec" in which:
            gen_codec()
        if "tapvid" in which:
            gen_tapvid()
        if "results" in which:
            gen_results_api()
        sys.exit(0)
    model = build_reference_model()
    if "ops" in which:
        gen_ops(model)
    if "flow" in which:
        gen_compute_flow(model)
    if "seq" in which:
        gen_chain_and_sequence()
    if "e2e" in which:
        gen_e2e(model)
    if "tapvid" in which:
        gen_tapvid()
    if "results" in which:
        gen_results_api()
    if "codec" in which:
        gen_codec()
