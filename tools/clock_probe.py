#!/usr/bin/env python3
"""The shader clock, MEASURED: every workgroup of the tile-resident conv kernel stamps s_memtime (shader cycles) and
s_memrealtime (the constant 100 MHz reference) when it starts and when it ends (tuning build with -DMFTX_LF_TRACE).
Per layer and grid size, for random and for all-zero operands (same binary -- the guide's power A/B):

    clock      = d(memtime) / d(memrealtime) x 100 MHz per workgroup (min / median / max)
    span       = last end - first start of the launch on the 100 MHz clock: the kernel's real duration
    skew       = spread of the workgroups' start times
    launch     = HIP events around a back-to-back loop of the same launch / its length: what a trace reports per launch

and the board's power / sclk sampled from sysfs while one kernel loops.

    tools/build_tuning.sh -DMFTX_LF_TRACE;  MFTX_LIB=build_tune/libmftx_tune.so python tools/clock_probe.py"""
import ctypes as C
import glob
import statistics
import sys
import threading
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd import _lib, ops  # noqa: E402

lib = _lib.load()
fn = lib.mftx_debug_tc_clock
fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_ulonglong), C.c_int]
h = w = 64
LOOP = 40


def probe(name, cin, cout, kh, kw, P, zeros):
    M = P * h * w
    g = torch.Generator().manual_seed(0)
    x = torch.zeros(M, cin) if zeros else torch.randn(M, cin, generator=g)
    wt = torch.zeros(cout, cin, kh, kw) if zeros else torch.randn(cout, cin, kh, kw, generator=g) * 0.05
    xs = ops.split_activations(x.cuda())
    x1, x2 = (xs, None) if cin == 128 else (xs[:, :128].contiguous(), xs[:, 128:].contiguous())
    wtile = ops.pack_tile_conv_weights(ops.pack_conv_weight(wt.cuda()), cout, cin)
    b = torch.zeros(cout).cuda()
    out = torch.empty(M, cout, device="cuda")
    th = 8 if kh == 3 else (4 if kh == 1 else 32)
    n_wg = P * -(-h // th) * -(-w // (128 // th))
    run = lambda: ops.tile_conv2d(x1, wtile, b, P, h, w, cout, kh, kw, act="relu", x2=x2, out_split=True, out=out)  # noqa: E731
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(LOOP):
        run()
    e1.record()
    torch.cuda.synchronize()
    launch_us = e0.elapsed_time(e1) * 1e3 / LOOP
    buf = (C.c_ulonglong * (4 * n_wg))()
    assert fn(buf, n_wg) == 0
    rec = [(buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]) for i in range(n_wg)]
    clk = [(t1 - t0) / max(r1 - r0, 1) * 0.1 for t0, t1, r0, r1 in rec]            # GHz
    cyc = [t1 - t0 for t0, t1, _, _ in rec]
    r_first, r_last = min(r[2] for r in rec), max(r[3] for r in rec)
    skew = (max(r[2] for r in rec) - r_first) * 0.01
    span = (r_last - r_first) * 0.01
    gf = 2.0 * M * cout * kh * kw * cin * 3e-9             # GFLOP of fp16 MFMA work (three products per fp32 product)
    print(f"{name:22s} P={P} wgs={n_wg:3d} {'zeros ' if zeros else 'random'} | launch {launch_us:6.1f} us  span {span:6.1f} us  start skew {skew:4.1f} us | "
          f"cycles/wg {statistics.median(cyc) / 1e3:6.1f} k | clock {min(clk):.2f} / {statistics.median(clk):.2f} / {max(clk):.2f} GHz | "
          f"{gf / launch_us * 1e3:5.0f} TF fp16 by launch, {gf / span * 1e3:5.0f} by span", flush=True)
    return run


def hwmon_files():
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        p = [f for f in (d + "/power1_average", d + "/power1_input") if Path(f).exists()]
        f = d + "/freq1_input"
        if p:
            out.append((p[0], f if Path(f).exists() else None))
    return out


def sysfs_sampler(stop, rows):
    """(t, power W, sclk GHz) of the board that draws the most: the box has many DRM cards, the busy one is ours."""
    files = hwmon_files()
    if not files:
        rows.append(None)
        return
    def read(path):
        try:
            return int(open(path).read())
        except (OSError, ValueError):
            return None
    # pick the card by one probe round taken while the GPU is already busy
    time.sleep(0.3)
    probe = [(read(p) or 0) for p, _ in files]
    pfile, ffile = files[max(range(len(files)), key=lambda i: probe[i])]
    while not stop.is_set():
        t = time.perf_counter()
        p, f = read(pfile), (read(ffile) if ffile else None)
        rows.append((t, p * 1e-6 if p is not None else None, f * 1e-9 if f is not None else None))
        time.sleep(0.002)


def power_run(name, run, seconds=2.5):
    stop, rows = threading.Event(), []
    th = threading.Thread(target=sysfs_sampler, args=(stop, rows))
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(200):
            run()
        torch.cuda.synchronize()
        n += 200
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    rows = [r for r in rows if r is not None and r[0] - t0 > 0.5]
    if not rows:
        print(f"{name}: no hwmon power / frequency files on this box ({n / dt * 1e-3:.1f} k launches/s)")
        return
    ps = [r[1] for r in rows if r[1] is not None]
    fs = [r[2] for r in rows if r[2] is not None]
    rate = len(rows) / max(rows[-1][0] - rows[0][0], 1e-9)
    msg = f"{name}: {n / dt * 1e-3:.1f} k launches/s = {dt / n * 1e6:.1f} us each, {len(rows)} samples at {rate:.0f} Hz"
    if ps: msg += f" | power {min(ps):.0f} / {statistics.median(ps):.0f} / {max(ps):.0f} W"
    if fs: msg += f" | sclk {min(fs):.2f} / {statistics.median(fs):.2f} / {max(fs):.2f} GHz"
    print(msg, flush=True)


print("# tools/clock_probe.py -- every workgroup's own first / last stamps of s_memtime and s_memrealtime (100 MHz); min / median / max over the workgroups of the LAST of", LOOP, "back-to-back launches")
runs = {}
for name, cin, cout, kh, kw in (("fh1 3x3 128->256", 128, 256, 3, 3), ("gru zr 1x5 256->256", 256, 256, 1, 5), ("gru q 1x5 256->128", 256, 128, 1, 5)):
    for P in ((1, 2, 4, 7) if kh == 3 else (1, 7)):
        for zeros in (False, True):
            r = probe(name, cin, cout, kh, kw, P, zeros)
            if P == 7: runs[(name, zeros)] = r
print("# power and sclk from sysfs (hwmon) while ONE kernel loops, 7 pairs")
for (name, zeros), r in runs.items():
    power_run(f"{name} {'zeros' if zeros else 'random'}", r)

# ---- inside the engine: the LAST tile-resident launch of a refinement is the mask head's first layer (3 x 3, 128 -> 256, the
# flow head's geometry) -- its stamps after ~5.5 ms of back-to-back GEMM work per call, production launch geometry, graph replay
from mft_amd.config import load_config  # noqa: E402
from mft_amd.synth import SyntheticVideo  # noqa: E402

REPO = Path(__file__).resolve().parents[1]
conf = load_config(REPO / "configs" / "MFT_cfg.py")
fc = conf.flow_config
fc.model, fc.synthetic_weights_seed, fc.flow_iters, fc.async_encode, fc.split_streams = None, 0, 12, False, 1
flower = fc.of_class(fc)
vid = SyntheticVideo(512, 512, n_frames=8, seed=0)
frames = [vid[i] for i in range(8)]
torch.cuda.set_stream(torch.cuda.Stream())
lefts, right = [(i, frames[i]) for i in range(7)], (7, frames[7])
for _ in range(6):
    flower.compute_flow_many(lefts, right)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    flower.compute_flow_many(lefts, right)
e1.record()
torch.cuda.synchronize()
n_wg = 224
buf = (C.c_ulonglong * (4 * n_wg))()
assert fn(buf, n_wg) == 0
rec = [(buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]) for i in range(n_wg)]
clk = [(t1 - t0) / max(r1 - r0, 1) * 0.1 for t0, t1, r0, r1 in rec]
cyc = [t1 - t0 for t0, t1, _, _ in rec]
r_first = min(r[2] for r in rec)
print(f"# in the engine (refinement of 7 pairs, {e0.elapsed_time(e1) / 10:.2f} ms per call, graph replay): mask head layer 1 (3x3 128->256, 224 workgroups): "
      f"span {(max(r[3] for r in rec) - r_first) * 0.01:.1f} us, start skew {(max(r[2] for r in rec) - r_first) * 0.01:.1f} us, "
      f"cycles/wg {statistics.median(cyc) / 1e3:.1f} k, clock {min(clk):.2f} / {statistics.median(clk):.2f} / {max(clk):.2f} GHz")
stop, rows = threading.Event(), []
th = threading.Thread(target=sysfs_sampler, args=(stop, rows))
th.start()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 3.0:
    for _ in range(20):
        flower.compute_flow_many(lefts, right)
    torch.cuda.synchronize()
stop.set()
th.join()
rows = [r for r in rows if r is not None and r[0] - t0 > 0.5]
ps, fs = [r[1] for r in rows if r[1] is not None], [r[2] for r in rows if r[2] is not None]
if ps or fs:
    print("# engine loop, 7 pairs, sysfs:" + (f" power {min(ps):.0f} / {statistics.median(ps):.0f} / {max(ps):.0f} W" if ps else "") +
          (f" sclk {min(fs):.2f} / {statistics.median(fs):.2f} / {max(fs):.2f} GHz" if fs else "") + f" ({len(rows)} samples)")
else:
    print("# engine loop: no hwmon power / frequency files on this box")
