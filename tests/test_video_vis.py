"""video ingest (image directory / frame array, pinned upload ring) and the demo overlays."""
import numpy as np
import pytest
import torch

from mft_amd import video as vio
from mft_amd import vis
from mft_amd.results import FlowOUTrackingResult


def _frames(n=5, H=24, W=40, seed=0):
    rng = np.random.default_rng(seed)
    return [rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8) for _ in range(n)]


def test_capture_reads_png_directory_and_frame_array(tmp_path):
    frames = _frames()
    d = tmp_path / "seq"
    for i, f in enumerate(frames):
        vio.imwrite_bgr(d / f"{i:03d}.png", f)
    (d / "notes.txt").write_text("ignored")
    assert vio.get_video_length(d) == len(frames)
    for a, b in zip(vio.get_video_frames(d), frames):
        assert a.dtype == np.uint8 and np.array_equal(a, b)             # BGR in, BGR out (cv2.imwrite / imread round trip)
    cap = vio.GeneralVideoCapture(d, reverse=True)
    ok, first = cap.read()
    assert ok and np.array_equal(first, frames[-1]) and cap.frame_src == "004.png"
    np.save(tmp_path / "v.npy", np.stack(frames))
    got = list(vio.get_video_frames(tmp_path / "v.npy"))
    assert len(got) == len(frames) and all(np.array_equal(a, b) for a, b in zip(got, frames))
    with pytest.raises(ValueError):
        np.save(tmp_path / "bad.npy", np.zeros((3, 4, 5), np.uint8))
        vio.GeneralVideoCapture(tmp_path / "bad.npy")
    # RGBA edit images keep their alpha and depth (cv2.IMREAD_UNCHANGED)
    edit = np.random.default_rng(1).integers(0, 256, size=(8, 9, 4), dtype=np.uint8)
    vio.imwrite_bgr(tmp_path / "edit.png", edit)
    assert np.array_equal(vio.imread_unchanged(tmp_path / "edit.png"), edit)
    assert np.array_equal(vio.imread_bgr(tmp_path / "edit.png"), edit[..., :3])
    if vio._cv2() is None:
        (tmp_path / "clip.mp4").write_bytes(b"\x00")
        with pytest.raises(RuntimeError):
            vio.GeneralVideoCapture(tmp_path / "clip.mp4")


def test_gray_and_blend():
    img = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 30]]], np.uint8)      # B, G, R
    g = vis.to_gray_3ch(img)
    assert g.shape == img.shape and g[0, :, 0].tolist() == [29, 150, 76, 128] and np.array_equal(g[..., 0], g[..., 2])
    a = np.full((1, 4), 128, np.uint8)
    out = vis.blend_with_alpha_premult(np.full_like(img, 100), img, a)
    want = (100 + img.astype(np.float32) * (1 - 128 / 255.0)).clip(0, 255).astype(np.uint8)
    assert np.array_equal(out, want)


def test_draw_dots_and_edit():
    H, W = 40, 56
    frame = np.zeros((H, W, 3), np.uint8)
    coords = np.array([[10.0, 12.0], [30.4, 20.6], [-50.0, 3.0], [55.0, 39.0]], np.float32)
    occl = np.array([0.0, 0.9, 0.0, 0.2], np.float32)
    out = vis.draw_dots(frame, torch.from_numpy(coords), occl)
    assert tuple(out[12, 10]) == vis.RED and tuple(out[12, 13]) == vis.RED and tuple(out[12, 15]) == (0, 0, 0)
    assert tuple(out[21, 30]) == (0, 0, 0)                                  # occluded point: no dot
    assert tuple(out[39, 55]) == vis.RED and not frame.any()                 # clipped at the border, input untouched
    # edit overlay: with zero flow the edit lands where it was painted, on the gray frame
    rng = np.random.default_rng(2)
    frame = rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)
    edit = np.zeros((H, W, 4), np.uint8)
    edit[5:15, 8:20] = (40, 80, 120, 255)
    res = FlowOUTrackingResult.identity((H, W))
    res.occlusion[0, 5:8, 8:12] = 1.0                                        # part of the edit is occluded
    out = vis.draw_edit(frame, res, edit)
    gray = vis.to_gray_3ch(frame)
    assert tuple(out[10, 10]) == (40, 80, 120)
    assert np.array_equal(out[6, 9], gray[6, 9]) and np.array_equal(out[30, 30], gray[30, 30])
    q = vis.get_queries((H, W), 30)
    assert q.shape == (4, 2) and q[-1].tolist() == [30.0, 30.0]


@pytest.mark.gpu
def test_frame_ring_and_result_drain():
    frames = _frames(n=11, H=64, W=96, seed=3)
    seen = []
    for f in vio.FrameRing(frames, keep=4):
        assert not f.is_cuda and f.is_pinned() and f.dtype == torch.uint8
        seen.append(f.cuda(non_blocking=True))                     # what the flow plugin does with it
    torch.cuda.synchronize()
    assert len(seen) == len(frames) and all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(seen, frames))
    drain = vio.ResultDrain()
    rs = []
    for i in range(4):
        r = FlowOUTrackingResult(torch.full((2, 16, 16), float(i), device="cuda"), torch.zeros(1, 16, 16, device="cuda"),
                                 torch.full((1, 16, 16), 2.0 * i, device="cuda"), validate=False)
        rs.append(r)
        drain.submit(r)
    for i in range(4):
        f, o, s = drain.collect()
        assert not f.is_cuda and float(f.mean()) == i and float(s.mean()) == 2.0 * i
    assert len(drain) == 0
    for i in range(6):                               # the pinned ring is reused; copy=True detaches a result from it
        drain.submit(rs[i % 4])
        f, _, _ = drain.collect(copy=True)
        assert float(f.mean()) == i % 4


@pytest.mark.gpu
def test_demo_on_png_directory(tmp_path):
    """demo.py end to end on a small PNG frame directory with an RGBA edit: overlays written, dots move with the flow."""
    import subprocess
    import sys
    from pathlib import Path
    from mft_amd.synth import SyntheticVideo
    repo = Path(__file__).resolve().parents[1]
    vid = SyntheticVideo(128, 160, n_frames=4, seed=1)
    for i in range(4):
        vio.imwrite_bgr(tmp_path / "in" / f"{i:03d}.png", vid[i])
    edit = np.zeros((128, 160, 4), np.uint8)
    edit[30:50, 40:70] = (0, 255, 255, 200)
    vio.imwrite_bgr(tmp_path / "edit.png", edit)
    res = subprocess.run([sys.executable, str(repo / "demo.py"), "--video", str(tmp_path / "in"), "--edit", str(tmp_path / "edit.png"),
                          "--out", str(tmp_path / "out"), "--synthetic_weights_seed", "7", "--grid_spacing", "16"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    pts = sorted((tmp_path / "out" / "in_points").glob("*.png"))
    eds = sorted((tmp_path / "out" / "in_edit").glob("*.png"))
    assert len(pts) == 4 and len(eds) == 4
    first = vio.imread_bgr(pts[0])
    assert first.shape == (128, 160, 3) and (first == np.array(vis.RED, np.uint8)).all(-1).sum() > 100
