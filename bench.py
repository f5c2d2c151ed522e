#!/usr/bin/env python3
"""Benchmark of the MFT hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...                      (re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one tracked 512x512 frame with the reference configuration (deltas {inf,1,2,4,8,16,32},
12 RAFT iterations, occlusion threshold 0.02; BASELINE.json configs[1]) on a seeded synthetic video
with seeded synthetic weights (no dataset / checkpoint in this environment).

Steady state.  The seven deltas are all live only from frame 33 on (MFT/MFT.py:74-91: delta 32 reaches
back to frame 1 there), so whatever --warmup is, an UNTIMED pre-roll tracks frames until every warm-up
and timed frame has 7 flow pairs; the pair count is recorded inside the timed loop (min / mean / max in
`config.workload` and `pairs_per_frame`) and the run fails rather than mislabel a ramp as steady state.
(Pre-roll = frames 1..32 always, so warm-up and timed frames all carry 7 pairs.)

Frames are resident in HBM before the timed region and results stay on the device; the PCIe-inclusive
rate (numpy frames in, CPU results out) is reported separately as "host_io_fps".  For N > 1 the
(frame, delta) flow computations of a look-ahead window of frames are sharded over the ranks
(mft_amd/dist.py): strong scaling of one video, `value` = frames of that one video per second.

Rank 0 prints ONE JSON line (see DESIGN.md section "Measurement").
"""
import argparse
import ctypes as C
import json
import os
import re
import socket
import subprocess
import sys
import time
from pathlib import Path

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# HIP multiplexes its streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and a stream that waits for an event blocks the
# queue it shares: with the tracker's streams (caller, encoders / exchange, two lanes, the graph proxies) four are too few for the
# sharded path -- emulated rank 0 of 8: window mode 1081 -> 1173 frames/s, per-frame mode 325 -> 559 -- while one GPU alone does not
# care (178.5 / 179.4).  Read by the HIP runtime when it initialises: set before torch is imported (mft_amd/__init__.py does the same).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

# stdout must carry ONE JSON line and nothing else.  Native libraries print to fd 1 from C (the RCCL version banner, some
# of it only when the process exits, through descriptors they set up when they are loaded), so fd 1 is pointed at stderr
# BEFORE torch is imported and the JSON line goes to the saved real stdout at the end.
REAL_STDOUT = None
if __name__ == "__main__" and not {"-h", "--help"} & set(sys.argv[1:]):
    sys.stdout.flush()
    REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
F16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_f16 / _bf16, dense (no sparsity)
SPLIT_PRODUCTS = 3              # split arithmetic: fp16 MFMA products per fp32 product (hi hi + hi lo + lo hi)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec peak
CATS = ["corr_volume_gemm", "corr_pool", "corr_lookup", "conv_gemm", "convf1", "glue", "convex_upsample",
        "chain_select", "conv_small_n", "encoder_instnorm", "lookup_convc1_fused", "flow_branch_fused",
        "encoder_conv_gemm", "gru_half_fused"]
FLOP_CATS = {0, 3, 4, 8, 11, 12, 13}
GEMM_PARTS = ("conv_gemm", "gru_half_fused", "encoder_conv_gemm")    # every conv GEMM of the step: what `roofline` prices, as in earlier rounds
# kernel symbols booked under GEMM_PARTS (profile.h: PC_CONV_GEMM / PC_GRU_HALF / PC_ENC_GEMM): the counter file is averaged over the
# same set that `roofline.achieved` and `flops_per_launch` are
GEMM_KERNEL_NAMES = ("conv_gemm", "tile_conv_kernel", "tile_conv2p_kernel", "gru_half_kernel", "ou_head_kernel")
VALU_CATS = {4, 8}
FULL_PAIRS = 7                  # flow pairs per frame once every delta is live
FIRST_FULL_FRAME = 33           # first frame index with FULL_PAIRS pairs (forward tracking from frame 0)
_T0 = time.time()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.time() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def oracle_threads():
    """Threads for the CPU oracle legs.  The oracle is torch CPU ops on one 512x512 pair at a time
    (M = 4096 rows per GEMM): it stops scaling long before a 256-core host is full -- the sweep on the
    GPU box's 256-core host (profiles/r2_cpu_thread_sweep.txt, tools/cpu_thread_sweep.py) runs a pair in
    0.69 / 0.52 / 0.89 / 2.2 / 7.0 / 783 s at 8 / 16 / 32 / 64 / 128 / 256 threads -- so the default is
    min(host cores, 16), the fastest; MFT_ORACLE_THREADS overrides."""
    return max(1, min(host_cores(), int(os.environ.get("MFT_ORACLE_THREADS", "16"))))


def build_tracker(args, sharded):
    from mft_amd.config import load_config
    conf = load_config(REPO / "configs" / "MFT_cfg.py")
    conf.flow_config.model = None                   # no checkpoint in this environment: explicit opt-in to
    conf.flow_config.synthetic_weights_seed = 0     # seeded synthetic weights
    conf.flow_config.flow_iters = args.iters
    conf.flow_config.async_encode = not args.sync_encode
    if getattr(args, "frames_in_flight", None):
        conf.flow_config.frames_in_flight = int(args.frames_in_flight)   # A/B: overrides the plugin's default (mft_amd/raft.py)
    if getattr(args, "alternate_corr", False):
        conf.flow_config.raft_params.alternate_corr = True
    conf.flow_config.raft_params.arith = getattr(args, "arith", "split")
    opts = {}
    if getattr(args, "no_graphs", False):
        opts["graph"] = 0                     # plain launches instead of hipGraph replays (A/B)
    if getattr(args, "no_fused_lookup", False):
        opts["fuse_lookup"] = 0               # lookup and convc1 as two kernels (A/B)
    for kv in getattr(args, "engine_opt", None) or []:
        k, sep, v = kv.partition("=")
        if not sep or not v.lstrip("-").isdigit():
            raise SystemExit(f"--engine-opt {kv!r}: expected NAME=INT")
        opts[k] = int(v)                      # A/B of a scheduling option of the refinement engine (mft_amd.ops.RaftEngine.OPTIONS)
    if opts:
        conf.flow_config.raft_params.engine_options = opts
    conf.keep_result_on_device = True
    conf.delta_sharding = sharded
    return conf.tracker_class(conf), conf


def profile_pass(tracker, frames, first, steps, arith="split"):
    """Same kind of steps again (steady state, 7 pairs each) with HIP-event brackets around every kernel
    launch.  Frames are encoded on the main stream and the 7 pairs run as one batch on one stream here, so
    that every kernel is timed alone (in the timed region the encoders of frame t+1 overlap frame t on a side
    stream and the batch runs as two halves on two streams -- overlapping kernels would stretch the bracketed
    intervals; that overlap is why `ms_per_step` is a little below the sum of these per-kernel times)."""
    from mft_amd import _lib
    lib = _lib.load()
    enc_stream = getattr(tracker.flower, "_enc_stream", None)
    split = getattr(tracker.flower, "_split_streams", 1)
    fif = getattr(tracker.flower, "_fif", 1)
    tracker.flower._enc_stream = None
    tracker.flower._split_streams = 1          # (and the batch as one part on one stream, for the same reason)
    tracker.flower._fif = 1                    # (and one frame in flight: the timed region overlaps two frames' kernel chains on two streams)
    torch.cuda.synchronize()
    lib.mftx_profile_begin()
    pairs = []
    for i in range(first, first + steps):
        tracker.track(frames[i])
        pairs.append(len(tracker.last_pairs))
    tracker.flower._enc_stream = enc_stream
    tracker.flower._split_streams = split
    tracker.flower._fif = fif
    n = len(CATS)
    ms, work, cnt = (C.c_double * n)(), (C.c_double * n)(), (C.c_longlong * n)()
    _lib.check(lib.mftx_profile_end(ms, work, cnt, n), "mftx_profile_end")
    out = {}
    for i, name in enumerate(CATS):
        if cnt[i] == 0:
            continue
        t = ms[i] * 1e-3
        d = {"launches": int(cnt[i]), "avg_us": 1e6 * t / cnt[i], "total_ms_per_step": ms[i] / steps}
        if i in FLOP_CATS:
            d.update(unit="TFLOP/s", achieved=work[i] / t / 1e12, peak=FP32_MFMA_PEAK_TFLOPS,
                     bound="valu" if i in VALU_CATS else "mfma", work_per_launch=work[i] / cnt[i])
            if name in ("conv_gemm", "corr_volume_gemm", "flow_branch_fused", "encoder_conv_gemm", "gru_half_fused") and arith == "split":
                # the update block's GEMMs run every fp32 product as three fp16 MFMA products: the matrix work actually
                # executed is 3 x the algorithmic flops, priced against the fp16 MFMA peak; the algorithmic rate is kept
                # next to it (it may exceed the fp32 MFMA peak, which this path does not use)
                d.update(algorithmic_tflops=d["achieved"], achieved=SPLIT_PRODUCTS * d["achieved"],
                         peak=F16_MFMA_PEAK_TFLOPS, arithmetic="split fp16 x 3, fp32 accumulate")
        elif work[i] > 0:
            d.update(unit="GB/s", achieved=work[i] / t / 1e9, peak=HBM_PEAK_GBS, bound="hbm",
                     work_per_launch=work[i] / cnt[i])
        if "achieved" in d:
            d["frac"] = d["achieved"] / d["peak"]
        out[name] = d
    # every conv GEMM of the step as one line (update block + OU heads on the ring-buffered and the tile-resident kernels, the fused
    # GRU passes, the encoders' layers): the dominant kernel family `roofline` reports
    parts = [CATS.index(n) for n in GEMM_PARTS if cnt[CATS.index(n)]]
    if parts:
        t = sum(ms[i] for i in parts) * 1e-3
        w = sum(work[i] for i in parts)
        c = sum(cnt[i] for i in parts)
        mult, peak = (SPLIT_PRODUCTS, F16_MFMA_PEAK_TFLOPS) if arith == "split" else (1, FP32_MFMA_PEAK_TFLOPS)
        out["conv_gemm_all"] = {"launches": int(c), "avg_us": 1e6 * t / c, "total_ms_per_step": 1e3 * t / steps, "unit": "TFLOP/s",
                                "achieved": mult * w / t / 1e12, "algorithmic_tflops": w / t / 1e12, "peak": peak, "bound": "mfma",
                                "work_per_launch": w / c, "frac": mult * w / t / 1e12 / peak, "parts": list(GEMM_PARTS)}
    return out, pairs


def api_default_pass(args, host_frames, n):
    """The reference's literal loop on the SHIPPED configuration (configs/MFT_cfg.py untouched but for the stand-in weights --
    there is no checkpoint here -- and the iteration count of the command line): numpy frames in, `tracker.track(frame)` returning
    `meta.result` as a CPU FlowOUTrackingResult (MFT/MFT.py:145-148, demo.py:59-65), default torch threads, no helper classes.
    Two loops over steady-state frames (7 pairs each):
      kept:     results are collected and read after the loop (a runner that writes its outputs at the end; demo.py keeps them in
                a list) -- `meta.result` is a PendingHostResult, so the tracker runs ahead of the copies;
      consumed: every result's planes are read before the next track() call (demo.py's convert_to_point_tracking does): one
                host synchronisation per frame, as with the reference -- the frame's latency, nothing overlaps."""
    from mft_amd.config import load_config
    conf = load_config(REPO / "configs" / "MFT_cfg.py")
    conf.flow_config.model = None
    conf.flow_config.synthetic_weights_seed = 0
    conf.flow_config.flow_iters = args.iters
    tr = conf.tracker_class(conf)
    warm = FIRST_FULL_FRAME + 7
    assert len(host_frames) >= warm + 2 * n, (len(host_frames), warm, n)
    tr.init(host_frames[0])
    for i in range(1, warm):
        tr.track(host_frames[i]).result.flow
    out = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kept = [tr.track(host_frames[i]).result for i in range(warm, warm + n)]
    checksum = sum(float(r.flow[0, 0, 0]) for r in kept)          # every result read on the host (waits for the last copies)
    torch.cuda.synchronize()
    out["kept"] = n / (time.perf_counter() - t0)
    assert all(not r.flow.is_cuda for r in kept) and np.isfinite(checksum)
    assert len(tr.last_pairs) == FULL_PAIRS, tr.last_pairs
    del kept
    t0 = time.perf_counter()
    for i in range(warm + n, warm + 2 * n):
        r = tr.track(host_frames[i]).result
        checksum += float(r.flow[0, 0, 0]) + float(r.occlusion[0, -1, -1])
    torch.cuda.synchronize()
    out["consumed"] = n / (time.perf_counter() - t0)
    out["threads"] = torch.get_num_threads()
    out["frames_in_flight"] = int(getattr(tr.flower, "_fif", 1))
    del tr
    torch.cuda.empty_cache()
    return out


def rank_diagnostics(tracker, window, H, W, world, rank, backend):
    """What the first real multi-GPU run needs to be self-diagnosing: per rank the device, the hardware-queue setting and whether it
    could take effect, the collective backend, and the time of the job's two all-gathers ALONE at this window's payload sizes (HIP
    events on the current stream, 5 repeats behind 2 warm-ups) -- the wire time the emulation (--emulate-world) leaves out."""
    import mft_amd
    from mft_amd.dist import split_units
    h, w = -(-H // 8), -(-W // 8)
    slots_f = -(-window // world)
    slots_u = max(c for _, c in split_units(FULL_PAIRS * window, world))
    d = {"rank": rank, "device": torch.cuda.current_device(), "device_name": torch.cuda.get_device_name(),
         "backend": backend, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "hw_queues": dict(mft_amd.HW_QUEUES),
         "frames_in_flight": int(getattr(tracker.flower, "_fif", 1)),
         "sharder": dict(getattr(tracker.sharder, "stats", {}))}
    real_world = dist.get_world_size()
    fake = real_world != world                                    # --emulate-world: this process stands for rank 0 of `world`, there is no wire
    for name, numel in (("features_all_gather", slots_f * h * w * 512), ("flowou_all_gather", slots_u * H * W * 4)):
        if fake:
            d[name] = {"bytes_sent_per_rank": 4 * numel, "bytes_gathered": 4 * numel * world, "us": None, "note": "emulated world: no wire"}
            continue
        send = torch.zeros(numel, dtype=torch.float32, device="cuda")
        recv = torch.empty(world * numel, dtype=torch.float32, device="cuda")
        for _ in range(2):
            dist.all_gather_into_tensor(recv, send)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            dist.all_gather_into_tensor(recv, send)
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / 5
        d[name] = {"bytes_sent_per_rank": 4 * numel, "bytes_gathered": 4 * numel * world, "us": us,
                   "bytes_per_us_received": 4 * numel * (world - 1) / us if us > 0 else None}
    return d


def host_io_pass(tracker, host_frames, base, n_io, n_io_frames, IO_WARM):
    """PCIe-inclusive variant of the timed loop on the SAME tracker, right behind the timed region: every frame comes from HOST memory
    and every result goes back to it (what the reference's API does, MFT/utils/io.py:566-615 in, MFT/MFT.py:145-148 out).  Frames
    wait in pinned buffers (mft_amd.video.FrameRing) and are uploaded by a copy kernel on the encoders' stream; results leave through
    the same copy kernel into pinned buffers and are collected two frames late (ResultDrain) -- no SDMA copy on either side, so
    nothing serialises (profiles/r2_io_paths.txt was the study of the hipMemcpyAsync paths)."""
    from mft_amd.video import FrameRing, ResultDrain
    enc = getattr(tracker.flower, "_enc_stream", None)
    ring = FrameRing((host_frames[i] for i in range(base, base + n_io_frames)), keep=40,
                     streams=[enc] if enc is not None else None).prepare(host_frames[0].shape)
    drain = ResultDrain(depth=4, nonfinite_from=tracker).prepare(tracker.memory[tracker.current_frame_i]['result'])
    got = 0
    t0 = None
    # (the loop issues no CPU tensor math; torch's intra-op pool -- 128 threads on this host -- only adds wake-up and
    # spin noise to the host-side waits: tools/io_paths3.py, 101 vs 124 frames/s)
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        for n, frame in enumerate(ring):
            if n == IO_WARM:              # the first frames touch the pinned buffers for the first time (GPU-side address
                torch.cuda.synchronize()  # translation of fresh pinned pages): steady state starts behind them
                t0 = time.perf_counter()
                got = -len(drain)
            drain.submit(tracker.track(frame).result)
            if len(drain) > 2:
                out = drain.collect()
                assert not out[0].is_cuda
                got += 1
        while len(drain):
            drain.collect()
            got += 1
        torch.cuda.synchronize()
        assert got == n_io, (got, n_io)
        return n_io / (time.perf_counter() - t0)
    finally:
        torch.set_num_threads(threads)


def build_hash():
    """sha256 (first 16 hex digits) of the library this process runs: ties a counter file to the build it was measured on."""
    import hashlib
    from mft_amd import _lib
    path = Path(os.environ.get("MFTX_LIB") or (REPO / "mft_amd" / "libmftx.so"))
    return hashlib.sha256(path.read_bytes()).hexdigest()[:16] if path.exists() else None


def profiled_traffic(profiles_dir=None):
    """HBM bytes per conv-GEMM launch from rocprofv3 PMC passes on this same command (tools/gpu_profile.sh ->
    profiles/*_pmc_hbm_traffic.csv: FETCH_SIZE x2-corrected + WRITE_SIZE in separate passes, launch-weighted).  Counters cannot be
    read in-process, so the file is tied to the build: its `# build:` line must name the library this run loads -- a file measured
    on another build is refused (traffic null, the reason in traffic_source)."""
    files = sorted(Path(profiles_dir or REPO / "profiles").glob("*_pmc_hbm_traffic.csv"))
    if not files:
        return None, None
    text = files[-1].read_text()
    m = re.search(r"^# build: (\w+)", text, re.M)
    mine = build_hash()
    if m is None or mine is None or m.group(1) != mine:
        return None, f"{files[-1].name} refused: measured on build {m.group(1) if m else 'unknown'}, this run loads {mine}"
    tot = n = 0.0
    for line in text.splitlines():
        if line.startswith("#") or not any(k in line for k in GEMM_KERNEL_NAMES):
            continue
        name, launches, _fetch, fetch_x2, write = line.rsplit(",", 4)     # the kernel name contains commas
        if "volume" in name or "pack_" in name or re.search(r"<\d+, \d+, \d+, \d+, 4[,>]", name):
            continue                                 # the correlation volume GEMM is its own category
        tot += float(launches) * (float(fetch_x2) + float(write)) * 1e6
        n += float(launches)
    return (tot / n if n else None), f"{files[-1].name} (build {mine})"


def cpu_baseline(args, vid):
    """The oracle (CPU restatement of the reference algorithm, torch CPU ops,
    features recomputed per pair like the reference) on a bounded sample."""
    from oracle import mft_oracle as O
    from mft_amd.weights import make_weights
    cores = oracle_threads()
    torch.set_num_threads(cores)
    sd = {k: torch.from_numpy(v) for k, v in make_weights(0).items()}
    n = args.cpu_frames
    stages = {}
    tr = O.Tracker(lambda l, r, li, ri: O.compute_flow(sd, li, ri, args.iters, timers=stages), timers=stages)
    tr.init(vid[0])
    H, W = vid[0].shape[:2]
    ident = (torch.zeros(2, H, W), torch.zeros(1, H, W), torch.zeros(1, H, W))
    for i in range(1, 33):              # steady-state memory ring without paying for 32 frames
        tr.memory[i] = dict(img=vid[i], result=ident)
    tr.cur = 32
    first_meta = None
    per_frame = []
    with torch.no_grad():
        t0 = time.perf_counter()
        for i in range(33, 33 + n):
            t1 = time.perf_counter()
            meta = tr.track(vid[i])
            per_frame.append(time.perf_counter() - t1)
            assert len(meta.pairs) == FULL_PAIRS
            first_meta = first_meta or meta
        dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": cores, "host_cores": host_cores(), "kind": "port",
            "sample": f"{n} steady-state frames (7 flow pairs x {args.iters} iters + chain + select each) of the same "
                      f"{H}x{W} synthetic video; oracle/mft_oracle.py on torch CPU ops, {cores} threads "
                      f"(thread sweep: profiles/r2_cpu_thread_sweep.txt), {dt:.1f} s.  The sample is bounded to the ~10-30 s of CPU "
                      f"work the bench contract allows (a frame takes ~3.5 s: 20 frames would be ~70 s; --cpu-frames 20 runs them, "
                      f"profiles/r6*_cpu_baseline_20.json); every frame is a steady-state frame (memory ring pre-filled), frame_s "
                      f"lists them",
            "s_per_frame": dt / n, "frame_s": [round(x, 3) for x in per_frame],
            "stage_s_per_frame": {k: v / n for k, v in stages.items()}}, first_meta


def track_parity(args, vid, oracle_meta):
    """One full MFT.track() step of the HIP path from the same state the cpu_baseline leg gave the
    oracle (frames 1..32 in memory with identity results, frame 33 arriving: 7 flow pairs, 7 chains,
    selection) against the oracle's result for that frame -- the oracle work is the baseline's."""
    from mft_amd.results import FlowOUTrackingResult
    tracker, _ = build_tracker(args, sharded=False)
    tracker.init(vid[0])
    H, W = vid[0].shape[:2]
    for i in range(1, 33):
        tracker.memory[i] = {"img": vid[i], "result": FlowOUTrackingResult.identity((H, W), device="cuda")}
    tracker.current_frame_i = 32
    got = tracker.track(vid[33]).result
    flow, occl, sigma = (t.cpu() for t in (got.flow, got.occlusion, got.sigma))
    rf, ro, rs = oracle_meta.result
    same = (tracker.last_chosen.cpu().long() == oracle_meta.chosen.long())
    epe = (flow - rf).pow(2).sum(0).sqrt()
    return {"track_flow_epe_px": epe.mean().item(),
            "track_flow_epe_px_where_same_delta": epe[same].mean().item(),
            "track_chosen_delta_agreement": same.float().mean().item(),
            "track_occlusion_max_abs_where_same_delta": (occl - ro).abs()[0][same].max().item(),
            "track_sigma_max_rel_where_same_delta": ((sigma - rs).abs() / rs.clamp_min(1e-6))[0][same].max().item()}


def track_parity_real_state(args, vid, n_frames=6):
    """Full-size parity with a REAL tracker state: `n_frames` consecutive frames from init with deltas {inf, 1, 2, 4} on the HIP
    tracker and on the oracle tracker -- from frame 2 on every chain samples a non-trivial `memory` result at fractional positions
    (the state of `track_parity` is the identity).  -> worst frame of the sequence."""
    from oracle import mft_oracle as O
    from mft_amd.weights import make_weights
    torch.set_num_threads(oracle_threads())
    sd = {k: torch.from_numpy(v) for k, v in make_weights(0).items()}
    deltas = [np.inf, 1, 2, 4]
    targs = argparse.Namespace(**vars(args))
    tracker, conf = build_tracker(targs, sharded=False)
    conf.deltas = list(deltas)
    tracker.init(vid[0])
    ref = O.Tracker(lambda l, r, li, ri: O.compute_flow(sd, li, ri, args.iters), deltas=deltas)
    ref.init(vid[0])
    worst = {"epe": 0.0, "epe_same": 0.0, "agree": 1.0, "occl": 0.0, "sigma": 0.0}
    pairs = 0
    t0 = time.perf_counter()
    for i in range(1, n_frames + 1):
        got = tracker.track(vid[i]).result
        with torch.no_grad():
            want = ref.track(vid[i])
        assert sorted(tracker.last_pairs) == sorted(want.pairs), (i, tracker.last_pairs, want.pairs)
        pairs += len(want.pairs)
        rf, ro, rs = want.result
        flow, occl, sigma = (t.cpu() for t in (got.flow, got.occlusion, got.sigma))
        same = tracker.last_chosen.cpu().long() == want.chosen.long()
        epe = (flow - rf).pow(2).sum(0).sqrt()
        worst["epe"] = max(worst["epe"], epe.mean().item())
        worst["epe_same"] = max(worst["epe_same"], epe[same].mean().item())
        worst["agree"] = min(worst["agree"], same.float().mean().item())
        worst["occl"] = max(worst["occl"], (occl - ro).abs()[0][same].max().item())
        worst["sigma"] = max(worst["sigma"], ((sigma - rs).abs() / rs.clamp_min(1e-6))[0][same].max().item())
    return {"real_state_frames": n_frames, "real_state_deltas": "inf,1,2,4", "real_state_pairs": pairs,
            "real_state_flow_epe_px_worst_frame": worst["epe"],
            "real_state_flow_epe_px_where_same_delta_worst_frame": worst["epe_same"],
            "real_state_chosen_delta_agreement_worst_frame": worst["agree"],
            "real_state_occlusion_max_abs_where_same_delta": worst["occl"],
            "real_state_sigma_max_rel_where_same_delta": worst["sigma"],
            "real_state_oracle_seconds": time.perf_counter() - t0}


def flow_epe_vs_oracle(args, tracker, vid):
    """EPE of one full-size flow pair (HIP path vs CPU oracle)."""
    from oracle import mft_oracle as O
    from mft_amd.weights import make_weights
    sd = {k: torch.from_numpy(v) for k, v in make_weights(0).items()}
    with torch.no_grad():
        ref = O.compute_flow(sd, vid[0], vid[4], args.iters)
    flow, extra = tracker.flower.compute_flow(vid[0], vid[4], mode="flow")
    epe = (flow.cpu() - ref[0]).pow(2).sum(0).sqrt().mean().item()
    return {"flow_epe_px": epe, "occlusion_max_abs": (extra["occlusion"].cpu() - ref[1]).abs().max().item(),
            "sigma_max_rel": ((extra["sigma"].cpu() - ref[2]).abs() / ref[2]).max().item()}


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def respawn_under_torchrun(n):
    """`python bench.py --gpus N` started as ONE process: run the same command as N ranks."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    log(f"--gpus {n} without a launcher: re-executing as {' '.join(cmd[1:8])} ...")
    return subprocess.run(cmd, env=env, stdout=REAL_STDOUT if REAL_STDOUT is not None else None).returncode


HOST_MS = []          # host time of every track() call (enqueue only: nothing waits for the GPU)


def run_frames(tracker, frames, first, count, window):
    """Track frames[first : first + count]; with `window` > 1 in look-ahead windows (multi-GPU).
    -> pairs per frame."""
    pairs = []
    if window <= 1:
        for i in range(first, first + count):
            t0 = time.perf_counter()
            tracker.track(frames[i])
            HOST_MS.append(1e3 * (time.perf_counter() - t0))
            pairs.append(len(tracker.last_pairs))
        return pairs
    # full windows of `window` frames, the remainder last (20 frames, window 8: 8 + 8 + 4): with window = world size every rank
    # gets exactly one full batch of 7 units per full window (8 x 7 units on 8 ranks); equal windows (7 + 7 + 6) round 49 / 8
    # units up to 7 per rank in every window -- measured by emulation at G = 8: 916 vs 858 frames/s.
    # MFT_BENCH_EQUAL_WINDOWS=1: the equal split.
    n_win = -(-count // window)
    sizes = [window] * (count // window) + ([count % window] if count % window else [])
    if os.environ.get("MFT_BENCH_EQUAL_WINDOWS"):
        sizes = [count // n_win + (1 if k < count % n_win else 0) for k in range(n_win)]
    i, got = first, 0
    for k, n in enumerate(sizes):
        nn = sizes[k + 1] if k + 1 < n_win else 0
        nxt = frames[i + n: i + n + nn]                              # the following window: its encoders start early
        # pipelined: the call hands back the PREVIOUS window's results; this window's result exchange and selections
        # overlap the next window's flow batches
        got += len(tracker.track_window(frames[i: i + n], next_imgs=nxt, defer=True))
        pairs += [len(tracker._plan(k)) for k in range(i, i + n)]
        i += n
    got += len(tracker.flush_window())
    assert got == count, (got, count)
    return pairs


def main():
    import faulthandler
    faulthandler.dump_traceback_later(240, repeat=True, file=sys.stderr)   # show where a stuck run is stuck
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--window", type=int, default=0,
                    help="multi-GPU look-ahead window in frames (0 = one per GPU; 1 = per-frame delta sharding)")
    ap.add_argument("--cpu-frames", type=int, default=6,
                    help="steady-state frames the CPU oracle is timed on (~3.5 s each on 16 threads; the default keeps the leg within "
                         "the ~10-30 s the bench contract allows)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the full-size EPE check against the CPU oracle")
    ap.add_argument("--no-host-io", action="store_true")
    ap.add_argument("--sync-encode", action="store_true", help="encode frames on the main stream")
    ap.add_argument("--frames-in-flight", type=int, default=None,
                    help="A/B: flow_config.frames_in_flight (consecutive frames' flow batches on that many engines / HIP streams)")
    ap.add_argument("--alternate-corr", action="store_true",
                    help="on-demand correlation (raft_params.alternate_corr): no stored volume, for memory-bound sizes")
    ap.add_argument("--arith", choices=("split", "fp32"), default="split",
                    help="raft_params.arith: split-fp16 products on the fp16 matrix cores (default) or fp32 MFMA")
    ap.add_argument("--no-alt-arith", action="store_true", help="skip the short pass in the other arithmetic")
    ap.add_argument("--no-graphs", action="store_true", help="A/B: plain kernel launches instead of hipGraph replays")
    ap.add_argument("--no-fused-lookup", action="store_true", help="A/B: correlation lookup and convc1 as two kernels")
    ap.add_argument("--engine-opt", action="append", metavar="NAME=INT",
                    help="A/B: a scheduling option of the refinement engine, e.g. fork=0 (results do not depend on it)")
    ap.add_argument("--ab-skip-encoders", action="store_true",
                    help="A/B upper bound only (the line is marked invalid): every frame reuses the first frame's features, no "
                         "encoder launches -- what the frame would cost if the encoders were free")
    ap.add_argument("--emulate-world", type=int, default=0,
                    help="with --force-sharded on ONE GPU: behave like rank 0 of this many ranks (compute only that rank's "
                         "share of every window; the other ranks' slots of the all-gathers are filled with copies of the "
                         "own data).  Results are meaningless; the time per window is what one rank of that world would "
                         "spend, without the wire time of the collectives: the projected multi-GPU rate")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the multi-GPU code path (windows, RCCL all-gathers) even with one rank (testing)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(respawn_under_torchrun(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # (tests only: MFT_DIST_ONE_GPU=1 puts every rank on cuda:0 and MFT_DIST_BACKEND=gloo carries the collectives on device tensors --
    # RCCL refuses two ranks on one device -- so that the N > 1 code path of this file runs as real processes on a one-GPU box;
    # the rates of such a run mean nothing)
    one_gpu = os.environ.get("MFT_DIST_ONE_GPU") == "1"
    backend = os.environ.get("MFT_DIST_BACKEND", "nccl")
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    sharded = world > 1 or args.force_sharded
    if sharded:
        if "MASTER_ADDR" not in os.environ:          # --force-sharded started without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0", WORLD_SIZE="1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    # default window = one frame per rank: 7 units per rank and window (one full batch), one frame to encode per rank
    # and window, and short windows pipeline well inside a 20-step timed region (emulated 8 ranks, --steps 20
    # --warmup 5: 671 frames/s with windows of 8 frames, 630 with 16, 610 with 24)
    window = (args.window if args.window > 0 else world) if sharded else 1

    from mft_amd.synth import SyntheticVideo
    preroll = FIRST_FULL_FRAME - 1                                # untimed: frames 1 .. 32; warm-up starts at frame 33
    n_io = 0 if (args.no_host_io or world > 1 or args.force_sharded) else max(8, args.steps)
    IO_WARM = 8                                                   # untimed frames in front of the host-io pass
    n_io_frames = n_io + (IO_WARM if n_io else 0)
    n_prof = 0 if args.no_profile else args.steps
    n_frames = 1 + preroll + args.warmup + args.steps + n_prof + n_io_frames
    n_api = 0 if (args.no_host_io or world > 1 or args.force_sharded) else max(20, args.steps)
    if n_api:
        n_frames = max(n_frames, FIRST_FULL_FRAME + 7 + 2 * n_api)
    vid = SyntheticVideo(args.height, args.width, n_frames=max(n_frames, 48), seed=0)
    host_frames = [vid[i] for i in range(n_frames)]
    frames = [torch.from_numpy(f).cuda() for f in host_frames]      # resident in HBM

    log(f"{n_frames} frames resident in HBM; host cores {host_cores()} (cpu_count {os.cpu_count()}); "
        f"pre-roll {preroll} + warm-up {args.warmup} + {args.steps} timed frames")
    tracker, conf = build_tracker(args, sharded=("force" if args.force_sharded and world == 1 else sharded))
    tracker.init(frames[0])
    if args.ab_skip_encoders:
        feats0 = tracker.flower._encode(frames[0])
        tracker.flower._encode = lambda img, _f=feats0: _f
    if args.emulate_world > 1:
        if not (args.force_sharded and world == 1):
            raise SystemExit("--emulate-world needs --force-sharded on one GPU")
        G = args.emulate_world
        sh = tracker.sharder
        sh.world_size, sh.rank = G, 0
        window = args.window if args.window > 0 else G

        def fake_all_gather(recv, send, group=None, async_op=False):      # every rank's slot := this rank's data
            recv.view(G, *send.shape).copy_(send.unsqueeze(0).expand(G, *send.shape))
            return None
        import mft_amd.dist as mdist
        mdist.dist = type("FakeDist", (), {"all_gather_into_tensor": staticmethod(fake_all_gather),
                                           "is_initialized": staticmethod(lambda: True)})
    torch.cuda.synchronize()
    t_ramp = time.perf_counter()
    ramp_pairs = run_frames(tracker, frames, 1, preroll, window)
    torch.cuda.synchronize()
    t_ramp = time.perf_counter() - t_ramp      # frames 1..preroll: 1 -> 7 flow pairs per frame, first-use set-up included
    warm_pairs = run_frames(tracker, frames, 1 + preroll, args.warmup, window)

    def fence():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
            torch.cuda.synchronize()

    first = 1 + preroll + args.warmup
    fence()
    del HOST_MS[:]
    t0 = time.perf_counter()
    timed_pairs = run_frames(tracker, frames, first, args.steps, window)
    host_ms = list(HOST_MS)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    log(f"timed region: {args.steps} steps in {dt:.3f} s, pairs/frame min {min(timed_pairs)} max {max(timed_pairs)}")
    if min(timed_pairs + warm_pairs) != FULL_PAIRS or max(timed_pairs) != FULL_PAIRS:
        raise SystemExit(f"timed region is not the 7-pair steady state: warm-up {warm_pairs}, timed {timed_pairs}")

    host_io_fps = None
    if n_io and rank == 0:
        host_io_fps = host_io_pass(tracker, host_frames, first + args.steps, n_io, n_io_frames, IO_WARM)
        log(f"host-io pass done: {host_io_fps:.1f} frames/s")
    api = None
    if n_api and rank == 0:
        # (before the per-kernel profile pass and before anything touches torch's thread count: the literal API loop with the process's
        # default threads, on a fresh tracker of the shipped configuration)
        api = api_default_pass(args, host_frames, n_api)
        log(f"api default pass: kept {api['kept']:.1f}, consumed each frame {api['consumed']:.1f} frames/s")
    kernels, prof_pairs = {}, []
    ranks_seen = 1
    rank_diag = None
    if sharded:
        # what the collectives themselves say about the job: every rank contributes a one
        ones = torch.ones(1, device="cuda", dtype=torch.float32)
        dist.all_reduce(ones)
        ranks_seen = int(round(float(ones.item())))
        mine = rank_diagnostics(tracker, window, args.height, args.width, args.emulate_world or world, rank, backend)
        print("[bench rank %d] %s" % (rank, json.dumps(mine)), file=sys.stderr, flush=True)
        if world > 1:
            rank_diag = [None] * world
            dist.all_gather_object(rank_diag, mine)
        else:
            rank_diag = [mine]
    if n_prof and not sharded:
        kernels, prof_pairs = profile_pass(tracker, frames, first + args.steps + n_io_frames, n_prof, args.arith)
        torch.cuda.synchronize()
        if min(prof_pairs) != FULL_PAIRS:
            raise SystemExit(f"profile pass left the steady state: {prof_pairs}")
    elif n_prof and rank == 0:
        # N > 1 (or forced sharding): the per-kernel roofline of the same workload, measured on rank 0 with a local,
        # unsharded tracker after the timed region (the other ranks wait at the final barrier) -- a rank of the sharded
        # job runs the very same kernels on batches of 7 or 8 pairs
        ptr, _ = build_tracker(args, sharded=False)
        ptr.init(frames[0])
        run_frames(ptr, frames, 1, preroll, 1)
        kernels, prof_pairs = profile_pass(ptr, frames, 1 + preroll, min(n_prof, 10), args.arith)
        torch.cuda.synchronize()
        del ptr

    result = None
    if rank == 0:
        fps = args.steps / dt
        result = {
            "metric": "tracked frames/sec at 512x512, 12 RAFT iters; flow EPE vs reference",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "f32" if args.arith == "fp32" else
            "f32 (update-block products as split fp16 x 3 on the fp16 matrix cores, fp32 accumulation; error per product "
            "<= ~2^-23, see DESIGN.md section 4 and parity below)",
            "arith": args.arith, "data": "synthetic",
            "config": {"workload": f"MFT.track on synthetic {args.height}x{args.width} video, deltas "
                                   f"[inf,1,2,4,8,16,32], {min(timed_pairs)}/{np.mean(timed_pairs):.2f}/{max(timed_pairs)} "
                                   f"(min/mean/max) flow pairs per timed frame, {args.iters} RAFT iters, seeded "
                                   f"synthetic weights (BASELINE.json configs[1])",
                       "parallelism": "single GPU" if not sharded else
                       f"(frame, delta) units of a {window}-frame look-ahead window sharded x{world}, "
                       f"feature + FlowOU all-gather over RCCL, replicated chain/select",
                       "frames_resident_in_hbm": True, "preroll_frames": preroll,
                       "first_timed_frame": first,
                       **({"test_only": f"{world} ranks share ONE GPU over {backend}: code-path test, the rate means nothing"} if one_gpu and world > 1 else {})},
            "emulated_world": (args.emulate_world or None),
            **({"invalid_ab": "encoders skipped (--ab-skip-encoders): an upper bound, not a measurement"} if args.ab_skip_encoders else {}),
            "ranks_seen": ranks_seen,          # sum over ranks of 1, by all-reduce (1 without a process group)
            **({"ranks": rank_diag} if rank_diag is not None else {}),
            "host_enqueue_ms_per_step": (float(np.mean(host_ms)) if host_ms else None),
            "pairs_per_frame": {"warmup": warm_pairs, "timed_min": min(timed_pairs), "timed_max": max(timed_pairs),
                                "timed_mean": float(np.mean(timed_pairs)), "profile_pass_min": min(prof_pairs, default=None)},
            "ramp": {"frames": preroll, "fps": (preroll / t_ramp) if preroll else None,
                     "pairs": ramp_pairs,
                     "note": "untimed pre-roll, frames 1..P after init: the number of flow pairs grows from 1 to 7"},
        }
        dom = kernels.get("conv_gemm_all") or kernels.get("conv_gemm")
        if dom:
            traffic, source = profiled_traffic()
            split = args.arith == "split"
            result["roofline"] = {"kernel": "conv_gemm_kernel + tile_conv_kernel + tile_conv2p_kernel + gru_half_kernel + ou_head_kernel (%s implicit GEMM, ring-buffered and tile-resident: update block, "
                                            "OU heads, encoders -- kernels.conv_gemm_all)" %
                                            ("split-fp16 MFMA" if split else "fp32 MFMA"),
                                  "bound": "mfma", "achieved": dom["achieved"], "peak": dom["peak"],
                                  "unit": "TFLOP/s", "frac": dom["frac"], "traffic": traffic,
                                  "traffic_source": source,
                                  "avg_launch_us": dom["avg_us"], "flops_per_launch": dom["work_per_launch"]}
            # the per-category times are bracketed with every kernel ALONE on one stream (profile_pass); the timed region overlaps
            # the encoders of frame t+1 (side stream) with frame t, so their sum exceeds ms_per_step by the overlap
            serial = sum(v["total_ms_per_step"] for k, v in kernels.items() if k != "conv_gemm_all")
            result["kernels_serial_ms_per_step"] = serial
            result["overlap_ms_per_step"] = serial - result["ms_per_step"]
            fif = int(getattr(tracker.flower, "_fif", 1))
            result["frames_in_flight"] = fif
            result["overlap_note"] = ("timed region: the flow batches of consecutive frames alternate between %d engines on %d HIP streams "
                                      "(flow_config.frames_in_flight: frame t + 1 needs frame t's features only, chaining / selection "
                                      "wait on an event) and the encoders run on a side stream; the per-kernel times above are "
                                      "bracketed with every kernel ALONE on the chip" % (fif, fif))
            # the step's executed matrix work over the WHOLE step time (non-GEMM kernels included): what the overlap buys shows here,
            # not in `frac` (a kernel timed alone)
            result["roofline"]["step_frac"] = dom["achieved"] * dom["total_ms_per_step"] / result["ms_per_step"] / dom["peak"]
            result["roofline"]["step_frac_note"] = "executed conv-GEMM flops of a step / ms_per_step / peak: the family's share of the peak over the whole frame time"
            lk = kernels.get("lookup_convc1_fused") or kernels.get("corr_lookup")
            if lk:
                result["roofline"]["north_star_lookup"] = {
                    "target_frac_of_hbm": 0.8, "achieved_frac": lk["frac"], "achieved_GBps": lk["achieved"],
                    "kernel": "lookup_convc1_fused" if "lookup_convc1_fused" in kernels else "corr_lookup",
                    "standalone_corr_lookup_frac": kernels.get("corr_lookup", {}).get("frac"),
                    "note": "algorithmic lookup bytes (SURVEY 8d) / kernel time vs 8 TB/s; at 512x512 the pyramid is "
                            "Infinity-Cache resident; DESIGN.md section 4 'lookup' has the 1080p (HBM-resident) figure"}
            if split:
                result["roofline"]["frac_algorithmic"] = dom["algorithmic_tflops"] / F16_MFMA_PEAK_TFLOPS
                result["roofline"].update(
                    note="achieved = 3 x algorithmic flops (the fp16 MFMA products executed) against the dense fp16 MFMA peak",
                    algorithmic_tflops=dom["algorithmic_tflops"],
                    algorithmic_vs_fp32_mfma_peak=dom["algorithmic_tflops"] / FP32_MFMA_PEAK_TFLOPS)
            result["kernels"] = kernels
    if not sharded and rank == 0 and not args.no_alt_arith:
        # the same steady-state frames in the other arithmetic (a fresh tracker; short: 10 timed frames)
        other = "fp32" if args.arith == "split" else "split"
        alt_args = argparse.Namespace(**{**vars(args), "arith": other})
        alt, _ = build_tracker(alt_args, sharded=False)
        alt.init(frames[0])
        run_frames(alt, frames, 1, preroll + 3, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        alt_pairs = run_frames(alt, frames, preroll + 4, 10, 1)
        torch.cuda.synchronize()
        alt_dt = time.perf_counter() - t0
        result["other_arith"] = {"arith": other, "value": 10 / alt_dt, "unit": "frames/s", "ms_per_step": 100 * alt_dt,
                                 "pairs_per_frame_min": min(alt_pairs)}
        del alt
        torch.cuda.empty_cache()
        log(f"other arithmetic ({other}): {10 / alt_dt:.1f} frames/s")
    if not sharded and rank == 0:
        if api is not None:
            result["api_default_fps"] = api["kept"]
            result["api_default_consumed_each_frame_fps"] = api["consumed"]
            result["api_default_note"] = (
                "reference's literal loop on the shipped configs/MFT_cfg.py (stand-in weights; %d torch threads; %d frames in flight): numpy "
                "frame in, tracker.track(frame).result a CPU FlowOUTrackingResult out (a PendingHostResult: the first access of its planes "
                "waits for the copy).  api_default_fps: results read after the loop; ..._consumed_each_frame_fps: every result read "
                "before the next track() call, i.e. one host synchronisation per frame as with the reference" %
                (api["threads"], api["frames_in_flight"]))
        if host_io_fps is not None:
            result["host_io_fps"] = host_io_fps
            result["host_io_path"] = ("pinned frames in and pinned results out, both moved by a copy kernel (no SDMA queue); the same tracker, "
                                      "right behind the timed region")
        torch.set_num_threads(oracle_threads())
        if not args.no_parity:
            result["parity"] = flow_epe_vs_oracle(args, tracker, vid)
            result["parity"].update(track_parity_real_state(args, vid))
            log(f"parity vs oracle: {result.get('parity')}")
        if not args.no_cpu_baseline:
            result["cpu_baseline"], oracle_meta = cpu_baseline(args, vid)
            log("cpu baseline done")
            if not args.no_parity:
                result.setdefault("parity", {}).update(track_parity(args, vid, oracle_meta))
                log(f"track() parity vs oracle: {result['parity']}")
    if rank == 0:
        os.write(REAL_STDOUT if REAL_STDOUT is not None else 1, (json.dumps(result) + "\n").encode())
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
