#!/usr/bin/env python3
"""Wall time and per-category kernel time of one `RaftEngine.refine` call (all of
SURVEY 8a4-a12 for P flow pairs, encoders excluded: features are cached) as a
function of the pair count P -- P = 7 is the single-GPU tracker step, P = 1..4
are what one rank sees under delta-sharding on 8..2 GPUs.

    python tools/bench_pairs.py [--P 1 2 4 7] [--size 512] [--iters 12]
"""
import argparse
import ctypes as C
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from mft_amd import _lib  # noqa: E402
from mft_amd.config import load_config  # noqa: E402
from mft_amd.synth import SyntheticVideo  # noqa: E402

CATS = ["corr_volume", "corr_pool", "lookup", "conv_gemm", "convf1", "glue", "upsample", "chain", "conv_small", "enc_norm",
        "lookup_fused", "flow_fused", "enc_gemm", "gru_fused"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--P", type=int, nargs="+", default=[1, 2, 4, 7])
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--own-stream", action="store_true", help="run on a non-default stream (graph capture needs one)")
    ap.add_argument("--engine-opt", action="append", metavar="NAME=INT", help="engine option, e.g. tile_conv=2 (always) / 0 (never)")
    a = ap.parse_args()
    conf = load_config(REPO / "configs" / "MFT_cfg.py")
    fc = conf.flow_config
    fc.model = None
    fc.synthetic_weights_seed = 0
    fc.flow_iters = a.iters
    fc.async_encode = False
    fc.split_streams = 1          # one batch on one stream: per-kernel event times must not overlap
    if a.engine_opt:
        for kv in a.engine_opt:
            if "=" not in kv or not kv.partition("=")[2].lstrip("-").isdigit():
                ap.error(f"--engine-opt {kv!r}: expected NAME=INT")
        fc.raft_params.engine_options = {kv.partition("=")[0]: int(kv.partition("=")[2]) for kv in a.engine_opt}
    flower = fc.of_class(fc)
    vid = SyntheticVideo(a.size, a.size, n_frames=8, seed=0)
    frames = [vid[i] for i in range(8)]
    lib = _lib.load()
    if a.own_stream:
        torch.cuda.set_stream(torch.cuda.Stream())
    for P in a.P:
        lefts = [(i, frames[i]) for i in range(P)]
        right = (7, frames[7])
        for _ in range(3):
            flower.compute_flow_many(lefts, right)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            flower.compute_flow_many(lefts, right)
        e1.record()
        torch.cuda.synchronize()
        wall = e0.elapsed_time(e1) / a.reps
        lib.mftx_profile_begin()
        for _ in range(a.reps):
            flower.compute_flow_many(lefts, right)
        n = len(CATS)
        ms, work, cnt = (C.c_double * n)(), (C.c_double * n)(), (C.c_longlong * n)()
        _lib.check(lib.mftx_profile_end(ms, work, cnt, n), "mftx_profile_end")
        parts = "  ".join(f"{CATS[i]} {ms[i] / a.reps:.2f}" for i in range(n) if cnt[i])
        gemm_ms = ms[3] + ms[13]
        gemm_tf = (work[3] + work[13]) / (gemm_ms * 1e-3) / 1e12 if gemm_ms else 0.0
        launches = sum(cnt[i] for i in range(n)) // a.reps
        print(f"P={P}: wall {wall:.2f} ms ({wall / P:.2f} ms/pair)  kernels {sum(ms) / a.reps:.2f} ms in {launches} launches  "
              f"conv_gemm {gemm_tf:.1f} TFLOP/s\n      {parts}")


if __name__ == "__main__":
    main()
