#!/usr/bin/env python3
"""Kernel-level determinism under GPU contention (run several copies at once, see tools/race_probe.py): every tile-resident kernel
on fixed random inputs, P pairs of 64 x 64 cells, repeated; reports the number of distinct outputs per kernel."""
import argparse
import hashlib
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from mft_amd import ops  # noqa: E402


def digest(*ts):
    hsh = hashlib.sha1()
    for t in ts:
        hsh.update(t.detach().cpu().numpy().tobytes())
    return hsh.hexdigest()[:10]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--pairs", type=int, default=1)
    ap.add_argument("--tag", default="")
    ap.add_argument("--load-seconds", type=float, default=0.0)
    ap.add_argument("--poison", action="store_true", help="ONE process, no load: fill every CU's LDS and registers with varying patterns in front of each call (tools/micro/poison.so)")
    a = ap.parse_args()
    dev = "cuda"
    g = torch.Generator(device="cpu").manual_seed(3)
    P, h, w = a.pairs, 64, 64
    M = P * h * w

    def rnd(*shape, s=1.0):
        return (torch.randn(*shape, generator=g) * s).to(dev)
    x128 = ops.split_activations(rnd(M, 128))
    mo128 = ops.split_activations(rnd(M, 128))
    hf = rnd(M, 128)
    hs = ops.split_activations(hf)
    tests = {}
    for kh, kw, N, cin in ((3, 3, 256, 128), (1, 5, 256, 256), (5, 1, 128, 256), (1, 5, 256, 128), (5, 1, 128, 128)):
        wpk = ops.pack_conv_weight(rnd(N, cin, kh, kw, s=0.05))
        wt = ops.pack_tile_conv_weights(wpk, N, cin)
        bias = rnd(N, s=0.1)
        x2 = mo128 if cin == 256 else None
        tests[f"tile_conv {kh}x{kw} cin{cin} N{N} relu"] = (lambda wt=wt, bias=bias, N=N, kh=kh, kw=kw, x2=x2:
                                                            (ops.tile_conv2d(x128, wt, bias, P, h, w, N, kh, kw, act="relu", x2=x2),))
    w1 = ops.pack_tile_conv_weights(ops.pack_conv_weight(rnd(256, 128, 3, 3, s=0.05)), 256, 128)
    wproj = ops.pack_flow_head_weights(ops.pack_conv_weight(rnd(2, 256, 3, 3, s=0.05)))
    b1, b2 = rnd(256, s=0.1), rnd(4, s=0.1)[:2].contiguous()
    b2p = torch.zeros(4, device=dev); b2p[:2] = b2
    tests["flow_head"] = lambda: (ops.flow_head(x128, h, w, w1, b1, wproj, b2p),)
    for vertical in (False, True):
        kh, kw = (5, 1) if vertical else (1, 5)
        wzr = ops.pack_tile_conv_weights(ops.pack_conv_weight(rnd(256, 256, kh, kw, s=0.05)), 256, 256)
        wq = ops.pack_tile_conv_weights(ops.pack_conv_weight(rnd(128, 256, kh, kw, s=0.05)), 128, 256)
        pre_zr, pre_q = rnd(M, 256, s=0.3), rnd(M, 128, s=0.3)
        tests[f"gru_half {'vertical' if vertical else 'horizontal'}"] = (
            lambda wzr=wzr, wq=wq, pre_zr=pre_zr, pre_q=pre_q, vertical=vertical: ops.gru_half(hs, mo128, wzr, wq, pre_zr, pre_q, hf, P, h, w, vertical=vertical))
    a712 = ops.split_activations(rnd(M, 712))
    wou1 = ops.pack_conv_weight(rnd(256, 712, 3, 3, s=0.03))
    wou2 = ops.pack_conv_weight(rnd(3, 256, 3, 3, s=0.05))
    wtile, wp2 = ops.pack_ou_heads_weights(wou1, wou2)
    bo1, bo2 = rnd(256, s=0.1), rnd(4, s=0.1)
    tests["ou_heads"] = lambda: (ops.ou_heads(a712, h, w, wtile, bo1, wp2, bo2),)
    if a.load_seconds:
        # load generator: whole refinements back to back for that many seconds (start one or two of these beside the tester)
        import time
        from mft_amd.config import Config
        from mft_amd.raft import RAFTWrapper
        from mft_amd.synth import SyntheticVideo
        from mft_amd.weights import make_weights
        c = Config(); c.flow_iters = 12
        fl = RAFTWrapper(c, state_dict=make_weights(7))
        vid = SyntheticVideo(512, 512, n_frames=9, seed=9)
        pairs = [(i, vid[i], 8, vid[8]) for i in range(2)]
        t0 = time.time()
        n = 0
        while time.time() - t0 < a.load_seconds:
            for _ in range(20):
                fl.compute_pairs(pairs, packed_out=True, planar=False)
            torch.cuda.synchronize()
            n += 20
        print(f"{a.tag} load: {n} refinements", flush=True)
        return
    poison = None
    if a.poison:
        import ctypes
        lib = ctypes.CDLL(str(Path(__file__).resolve().parent / "micro" / "poison.so"))
        lib.poison_launch.argtypes = [ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p]
        sink = torch.zeros(4, dtype=torch.int32, device=dev)
        patterns = [0x7fc00000, 0xffffffff, 0x00000000, 0x7f800000, 0x3f800000, 0x7bff7bff, 0xff800000]

        def poison(i):
            assert lib.poison_launch(patterns[i % len(patterns)], sink.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    for name, fn in tests.items():
        sums = []
        for i in range(a.reps):
            if poison is not None:
                poison(i)
            out = fn()
            sums.append(torch.stack([t.view(torch.int32).to(torch.int64).sum() for t in out]).sum())     # on the device: no sync per call
        vals = torch.stack(sums).cpu().tolist()
        seen = {}
        for v in vals:
            seen[v] = seen.get(v, 0) + 1
        print(f"{a.tag} P={P} {name:34s} distinct {len(seen)} {sorted(seen.values(), reverse=True)[:6]}", flush=True)


if __name__ == "__main__":
    main()
