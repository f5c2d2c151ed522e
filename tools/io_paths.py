#!/usr/bin/env python3
"""Where the PCIe-inclusive loop spends its time: frames/s of the tracker with device-resident frames, frames through
the pinned upload ring, results through the pinned drain, both, and the plain numpy-in / .cpu()-out path, plus the host
time spent inside each call (a host-bound call starves the GPU).  Run on the GPU box."""
import sys
import time
from collections import defaultdict
from pathlib import Path
from types import SimpleNamespace

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from mft_amd.synth import SyntheticVideo  # noqa: E402
from mft_amd.video import FrameRing, ResultDrain  # noqa: E402

args = SimpleNamespace(iters=12, sync_encode=False)
N = 24
vid = SyntheticVideo(512, 512, n_frames=40 + 6 * N, seed=0)
host = [vid[i] for i in range(40 + 6 * N)]
dev = [torch.from_numpy(f).cuda() for f in host]
tr, conf = bench.build_tracker(args, False)
tr.init(dev[0])
for i in range(1, 40):
    tr.track(dev[i])
torch.cuda.synchronize()
pos = [40]
host_t = defaultdict(float)


class T:
    def __init__(self, k): self.k = k
    def __enter__(self): self.t = time.perf_counter()
    def __exit__(self, *a): host_t[self.k] += time.perf_counter() - self.t


def run(name, body):
    host_t.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    body(pos[0])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pos[0] += N
    parts = "  ".join(f"{k} {1e3 * v / N:.2f}" for k, v in host_t.items())
    print(f"{name:42s} {N / dt:6.1f} fps   host ms/frame: {parts}")


def dev_loop(p):
    for i in range(p, p + N):
        with T("track"): tr.track(dev[i])


def ring_loop(p):
    it = iter(FrameRing(host[p: p + N]))
    while True:
        with T("ring"):
            f = next(it, None)
        if f is None:
            break
        with T("track"): tr.track(f)


def drain_loop(p):
    d = ResultDrain()
    for i in range(p, p + N):
        with T("track"): m = tr.track(dev[i])
        with T("submit"): d.submit(m.result)
        with T("collect"):
            while len(d) > 2: d.collect()
    while len(d): d.collect()


def both_loop(p):
    d = ResultDrain()
    it = iter(FrameRing(host[p: p + N]))
    while True:
        with T("ring"):
            f = next(it, None)
        if f is None:
            break
        with T("track"): m = tr.track(f)
        with T("submit"): d.submit(m.result)
        with T("collect"):
            while len(d) > 2: d.collect()
    while len(d): d.collect()


def simple_loop(p):
    conf.keep_result_on_device = False
    for i in range(p, p + N):
        with T("track"): tr.track(host[i])
    conf.keep_result_on_device = True


def ring_sync_loop(p):
    conf.keep_result_on_device = False
    for f in FrameRing(host[p: p + N]):
        with T("track"): tr.track(f)
    conf.keep_result_on_device = True


def ring_drain1_loop(p):
    """one packed D2H per frame instead of three"""
    pinned = [torch.empty(4, 512, 512).pin_memory() for _ in range(4)]
    evs = []
    for k, f in enumerate(FrameRing(host[p: p + N])):
        with T("track"): m = tr.track(f)
        with T("submit"):
            r = m.result
            pinned[k % 4].copy_(torch.cat([r.flow, r.occlusion, r.sigma], 0), non_blocking=True)
            ev = torch.cuda.Event(); ev.record(); evs.append(ev)
        with T("collect"):
            if len(evs) > 2: evs.pop(0).synchronize()
    for e in evs: e.synchronize()


run("device frames, device results", dev_loop)
run("ring in, .cpu() out", ring_sync_loop)
run("ring in, one packed pinned copy out", ring_drain1_loop)
run("ring in, device results", ring_loop)
run("device frames, drain out", drain_loop)
run("ring in, drain out", both_loop)
run("numpy in, .cpu() out (round-1 path)", simple_loop)
run("device frames, device results (again)", dev_loop)
