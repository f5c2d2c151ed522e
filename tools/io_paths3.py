#!/usr/bin/env python3
"""Round-3 study of the PCIe-inclusive loop (frames from host memory, results to host memory) with the copy-kernel
transport: which combination of FrameRing / ResultDrain costs what.  512 x 512, 7 pairs per frame."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from mft_amd.config import load_config  # noqa: E402
from mft_amd.synth import SyntheticVideo  # noqa: E402
from mft_amd.video import FrameRing, ResultDrain  # noqa: E402

import os
UP, DOWN, GRAPH = os.environ.get("IO_UP", "kernel"), os.environ.get("IO_DOWN", "kernel"), int(os.environ.get("IO_GRAPH", "1"))
from mft_amd import ops, raft, video  # noqa: E402
if UP == "sdma":
    def _dev_img(self, img_bgr):
        img = img_bgr if isinstance(img_bgr, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(img_bgr))
        return img.to(self.device, non_blocking=True).contiguous()
    raft.RAFTWrapper._device_image = _dev_img
if DOWN != "kernel":
    _side = torch.cuda.Stream() if DOWN == "kernel_side" else None
    def _submit(self, result):
        planes = result.planes()
        slot = self._n % self.depth
        if len(self._sets) <= slot:
            self._sets.append([torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in planes])
        host = self._sets[slot]
        if _side is not None:
            _side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(_side):
                for h, t in zip(host, planes):
                    ops.copy_bytes(t.contiguous(), h)
                ev = torch.cuda.Event(); ev.record(_side)
            for t in planes:
                t.record_stream(_side)
        else:
            for h, t in zip(host, planes):
                h.copy_(t, non_blocking=True)
            ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream())
        self._queue.append((ev, host)); self._n += 1
    video.ResultDrain.submit = _submit
if os.environ.get("IO_THREADS"):
    torch.set_num_threads(int(os.environ["IO_THREADS"]))
print(f"== upload {UP}, download {DOWN}, graphs {GRAPH}, torch threads {torch.get_num_threads()}")
conf = load_config(REPO / "configs" / "MFT_cfg.py")
if not GRAPH:
    conf.flow_config.raft_params.engine_options = {"graph": 0}
conf.flow_config.model = None
conf.flow_config.synthetic_weights_seed = 0
conf.flow_config.async_encode = True
conf.keep_result_on_device = True
N0, NT = 36, 24
vid = SyntheticVideo(512, 512, n_frames=N0 + 4 * NT + 2, seed=0)
host = [vid[i] for i in range(N0 + 4 * NT + 2)]
dev = [torch.from_numpy(f).cuda() for f in host]
tr = conf.tracker_class(conf)
tr.init(dev[0])
for i in range(1, N0):
    tr.track(dev[i])
torch.cuda.synchronize()


def run(name, ring_in, drain_out, base):
    enc = tr.flower._enc_stream
    frames = (FrameRing((host[i] for i in range(base, base + NT)), keep=40, streams=[enc]).prepare(host[0].shape)
              if ring_in else (dev[i] for i in range(base, base + NT)))
    drain = ResultDrain(depth=4) if drain_out else None
    th = {"next": 0.0, "track": 0.0, "submit": 0.0, "collect": 0.0}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    it = iter(frames)
    while True:
        z = time.perf_counter()
        try:
            f = next(it)
        except StopIteration:
            break
        a = time.perf_counter()
        th["next"] += a - z
        res = tr.track(f).result
        b = time.perf_counter()
        th["track"] += b - a
        if drain is not None:
            drain.submit(res)
            c = time.perf_counter()
            th["submit"] += c - b
            if len(drain) > 2:
                drain.collect()
            th["collect"] += time.perf_counter() - c
    while drain is not None and len(drain):
        drain.collect()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:36s} {NT / dt:6.1f} fps   host ms/frame: " + "  ".join(f"{k} {1e3 * v / NT:.2f}" for k, v in th.items()))


if os.environ.get("IO_ALL"):
    run("device frames, device results", False, False, N0)
    run("ring in, device results", True, False, N0 + NT)
    run("device frames, drain out", False, True, N0 + 2 * NT)
run("ring in, drain out", True, True, N0 + 3 * NT)
