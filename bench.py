#!/usr/bin/env python3
"""Benchmark of the MFT hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 40
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one MFT.track() of a 512x512 frame with the reference configuration
(deltas {inf,1,2,4,8,16,32}, 12 RAFT iterations, occlusion threshold 0.02;
BASELINE.json configs[1]) on a seeded synthetic video with seeded synthetic
weights (no dataset / checkpoint in this environment).  Warm-up defaults to 40
frames so that all seven deltas are live in the timed region.  Frames are
resident in HBM before the timed region and results stay on the device; the
PCIe-inclusive rate (numpy frames in, CPU results out) is reported separately
as "host_io_fps".  For N > 1 the per-frame flow deltas are sharded over ranks
(one RCCL all-gather per frame), i.e. strong scaling of one video.

Rank 0 prints ONE JSON line (see DESIGN.md section "Measurement").
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec peak
CATS = ["corr_volume_gemm", "corr_pool", "corr_lookup", "conv_gemm", "convf1", "glue", "convex_upsample",
        "chain_select", "conv_small_n", "encoder_instnorm"]
FLOP_CATS = {0, 3, 4, 8}
VALU_CATS = {4, 8}
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
    """Threads for the CPU oracle legs: all host cores up to 32 (the torch CPU ops
    of this small-batch workload stop scaling well before that)."""
    return max(1, min(host_cores(), int(os.environ.get("MFT_ORACLE_THREADS", "32"))))


def build_tracker(args, sharded):
    from mft_amd.config import load_config
    conf = load_config(REPO / "configs" / "MFT_cfg.py")
    conf.flow_config.synthetic_weights_seed = 0
    conf.flow_config.flow_iters = args.iters
    conf.flow_config.async_encode = not args.sync_encode
    conf.flow_config.torch_encoders = args.torch_encoders
    conf.keep_result_on_device = True
    conf.delta_sharding = sharded
    return conf.tracker_class(conf), conf


def profile_pass(tracker, frames, first, steps):
    """Same steps again with HIP-event brackets around every kernel launch.  Frames are encoded on the
    main stream here so that every kernel is timed alone (in the timed region the encoders of frame
    t+1 overlap frame t on a side stream, which would stretch the bracketed intervals)."""
    from mft_amd import _lib
    lib = _lib.load()
    enc_stream = getattr(tracker.flower, "_enc_stream", None)
    tracker.flower._enc_stream = None
    torch.cuda.synchronize()
    lib.mftx_profile_begin()
    for i in range(first, first + steps):
        tracker.track(frames[i])
    tracker.flower._enc_stream = enc_stream
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
        elif work[i] > 0:
            d.update(unit="GB/s", achieved=work[i] / t / 1e9, peak=HBM_PEAK_GBS, bound="hbm",
                     work_per_launch=work[i] / cnt[i])
        if "achieved" in d:
            d["frac"] = d["achieved"] / d["peak"]
        out[name] = d
    return out


def profiled_traffic():
    """HBM bytes per conv-GEMM launch measured offline with rocprofv3 PMC passes on this same
    command (profiles/*_pmc_hbm_traffic.csv: FETCH_SIZE x2-corrected + WRITE_SIZE, launch-weighted)."""
    files = sorted((REPO / "profiles").glob("*_pmc_hbm_traffic.csv"))
    if not files:
        return None, None
    tot = n = 0.0
    for line in files[-1].read_text().splitlines():
        if line.startswith("#") or "conv_gemm_kernel" not in line:
            continue
        name, launches, _fetch, fetch_x2, write = line.rsplit(",", 4)     # the kernel name contains commas
        import re
        if "<128, 128," in name or re.search(r"<\d+, \d+, \d+, \d+, 4(, \d+)?>", name):   # correlation volume (EPI_VOLUME = 4; 128x128 before r1h)
            continue
        tot += float(launches) * (float(fetch_x2) + float(write)) * 1e6
        n += float(launches)
    return (tot / n if n else None), files[-1].name


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
    with torch.no_grad():
        t0 = time.perf_counter()
        for i in range(33, 33 + n):
            meta = tr.track(vid[i])
            assert len(meta.pairs) == 7
            first_meta = first_meta or meta
        dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n} steady-state frames (7 flow pairs x {args.iters} iters + chain + select each) of the same "
                      f"{H}x{W} synthetic video; oracle/mft_oracle.py on torch CPU ops, {dt:.1f} s",
            "s_per_frame": dt / n,
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


def main():
    import faulthandler
    faulthandler.dump_traceback_later(240, repeat=True, file=sys.stderr)   # show where a stuck run is stuck
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--cpu-frames", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the full-size EPE check against the CPU oracle")
    ap.add_argument("--sync-encode", action="store_true", help="encode frames on the main stream")
    ap.add_argument("--torch-encoders", action="store_true", help="PyTorch-ROCm/MIOpen encoders instead of the native ones")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from mft_amd.synth import SyntheticVideo
    n_frames = 1 + args.warmup + 2 * args.steps + 8
    vid = SyntheticVideo(args.height, args.width, n_frames=max(n_frames, 48), seed=0)
    host_frames = [vid[i] for i in range(n_frames)]
    frames = [torch.from_numpy(f).cuda() for f in host_frames]      # resident in HBM

    log(f"{n_frames} frames resident in HBM; host cores {host_cores()} (cpu_count {os.cpu_count()})")
    tracker, conf = build_tracker(args, sharded=world > 1)
    tracker.init(frames[0])
    torch.cuda.synchronize()
    t_ramp = time.perf_counter()
    for i in range(1, 1 + args.warmup):
        tracker.track(frames[i])
        if i in (1, 2, 8, 33):
            torch.cuda.synchronize()
            log(f"warm-up frame {i} done ({len(tracker.last_pairs)} pairs)")
    torch.cuda.synchronize()
    t_ramp = time.perf_counter() - t_ramp      # frames 1..W: 1 -> 7 flow pairs per frame, first-use set-up included

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    first = 1 + args.warmup
    fence()
    t0 = time.perf_counter()
    for i in range(first, first + args.steps):
        tracker.track(frames[i])
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    log(f"timed region: {args.steps} steps in {dt:.3f} s")
    kernels = {}
    if not args.no_profile:
        kernels = profile_pass(tracker, frames, first + args.steps, args.steps)
        torch.cuda.synchronize()

    result = None
    if rank == 0:
        fps = args.steps / dt
        K = len(tracker.last_pairs)
        result = {
            "metric": "tracked frames/sec at 512x512, 12 RAFT iters; flow EPE vs reference",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"MFT.track on synthetic {args.height}x{args.width} video, deltas "
                                   f"[inf,1,2,4,8,16,32] ({K} flow pairs/frame in the timed region), "
                                   f"{args.iters} RAFT iters, seeded synthetic weights (BASELINE.json configs[1])",
                       "parallelism": "single GPU" if world == 1 else f"delta-sharded x{world} + all-gather",
                       "frames_resident_in_hbm": True},
            "ramp": {"frames": args.warmup, "fps": args.warmup / t_ramp,
                     "note": "untimed warm-up, frames 1..W after init: the number of flow pairs grows from 1 to 7"},
        }
        dom = kernels.get("conv_gemm")
        if dom:
            result["roofline"] = {"kernel": "conv_gemm_kernel (fp32 MFMA implicit GEMM: update block + OU heads)",
                                  "bound": "mfma", "achieved": dom["achieved"], "peak": dom["peak"],
                                  "unit": "TFLOP/s", "frac": dom["frac"], "traffic": profiled_traffic()[0],
                                  "traffic_source": profiled_traffic()[1],
                                  "avg_launch_us": dom["avg_us"], "flops_per_launch": dom["work_per_launch"]}
            result["kernels"] = kernels
    if world == 1 and rank == 0:
        # PCIe-inclusive variant of the same loop: numpy frames in, CPU results out
        conf.keep_result_on_device = False
        n_io = max(4, args.steps // 2)
        base = first + 2 * args.steps
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(base, min(base + n_io, n_frames)):
            tracker.track(host_frames[i])
        torch.cuda.synchronize()
        result["host_io_fps"] = (min(base + n_io, n_frames) - base) / (time.perf_counter() - t0)
        log("host-io pass done")
        torch.set_num_threads(oracle_threads())
        if not args.no_parity:
            result["parity"] = flow_epe_vs_oracle(args, tracker, host_frames)
            log(f"parity vs oracle: {result.get('parity')}")
        if not args.no_cpu_baseline:
            result["cpu_baseline"], oracle_meta = cpu_baseline(args, host_frames)
            log("cpu baseline done")
            if not args.no_parity:
                result.setdefault("parity", {}).update(track_parity(args, host_frames, oracle_meta))
                log(f"track() parity vs oracle: {result['parity']}")
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
