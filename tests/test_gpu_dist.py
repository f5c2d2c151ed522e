"""Window-sharded tracking over RCCL on the GPUs that are visible (1 on the test box: the sharded
code path -- feature all-gather, share of the (frame, delta) units, FlowOU all-gather, replicated
chain + select -- is forced and must equal the plain single-GPU tracker bit for bit; with more GPUs
visible the same test runs one rank per GPU)."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parents[1]


@pytest.mark.timeout(600)
def test_window_sharding_rccl_bitwise(tmp_path):
    n = min(torch.cuda.device_count(), 4)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29621", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", "29621", str(REPO / "tests" / "gpu_dist_worker.py"),
           str(tmp_path)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=550)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    single = np.load(tmp_path / "single.npz")
    for mode in ("L1", "L6", "L6p", "L6d"):
        enc = 0.0
        for r in range(n):
            got = np.load(tmp_path / f"rank{r}_{mode}.npz")
            enc += float(got["_encoded"])
            for k in single.files:
                assert np.array_equal(got[k], single[k]), (mode, r, k)
        if mode in ("L6", "L6p", "L6d") and n <= 6:
            # every exchanged frame is encoded exactly once across the ranks (the ragged last window of one
            # frame is shorter than the world size when n > 1: every rank encodes it itself)
            assert enc == 13, enc


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_real_plugin_multirank_on_one_gpu(tmp_path, world):
    """The REAL plugin with world_size > 1 ("weak" 5a): `world` processes share cuda:0, the collectives run
    on device tensors through gloo (RCCL refuses two ranks on one device; the code path of mft_amd/dist.py is the same:
    encode_packed / encode_half / adopt_*, the side-stream exchange, record_stream, asynchronous all_gather_into_tensor +
    wait, engine-written send buffers).  Every rank, in the per-frame mode (L = 1: the frame's two encoders on two ranks),
    with one frame per rank (L = G), with look-ahead windows of 6 frames plain / prefetched / pipelined, with a warm flow
    cache, and on a 512 x 512 window at production settings, is BITWISE equal to the single-process tracker."""
    port = str(29630 + world)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0",
               MFT_DIST_BACKEND="gloo", MFT_DIST_ONE_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", port, str(REPO / "tests" / "gpu_dist_worker.py"), str(tmp_path)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=850)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    single = np.load(tmp_path / "single.npz")
    for mode in ("L1", "LG", "L6", "L6p", "L6d", "cachecold", "cachewarm"):
        enc = 0.0
        for r in range(world):
            got = np.load(tmp_path / f"rank{r}_{mode}.npz")
            enc += float(got["_encoded"]) if "_encoded" in got.files else 0.0
            for k in single.files:
                assert np.array_equal(got[k], single[k]), (mode, r, k)
        if not mode.startswith("cache"):
            assert enc == 13.0, (mode, enc)            # every frame encoded exactly once across the ranks, whatever the window
    warm = [np.load(tmp_path / f"rank{r}_cachewarm.npz") for r in range(world)]
    # warm run: every finite-delta unit comes from its owner's cache, nothing is written again
    assert sum(int(w["_hits"]) for w in warm) == sum(int(w["_cold_writes"]) for w in warm) > 0
    assert all(int(w["_warm_writes"]) == 0 for w in warm)
    for name in ("big", "big1"):
        ref = np.load(tmp_path / f"single_{name}.npz")
        for r in range(world):
            got = np.load(tmp_path / f"rank{r}_{name}.npz")
            for k in ref.files:
                assert np.array_equal(got[k], ref[k]), (name, r, k)


@pytest.mark.timeout(900)
def test_c5_1080p_windows_sharded_over_two_ranks_on_one_gpu(tmp_path):
    """BASELINE config 5 (1080p, the pyramid resident in HBM, multi-GPU): two real ranks on the one GPU of the test box (gloo on
    device tensors), look-ahead windows of two 1080 x 1920 frames, prefetched and pipelined, 12 iterations, deltas inf / 1 / 2 / 4 / 8:
    every rank bitwise equal to the single-process tracker (whose 1080p pairs are held against the oracle in test_gpu_e2e.py)."""
    port = "29655"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0",
               MFT_DIST_BACKEND="gloo", MFT_DIST_ONE_GPU="1", MFT_DIST_1080P="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", port, str(REPO / "tests" / "gpu_dist_worker.py"), str(tmp_path)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=850)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    ref = np.load(tmp_path / "single_hd.npz")
    assert ref["flow1"].shape == (2, 1080, 1920) and len([k for k in ref.files if k.startswith("flow")]) == 4
    for r in range(2):
        got = np.load(tmp_path / f"rank{r}_hd.npz")
        for k in ref.files:
            assert np.array_equal(got[k], ref[k]), (r, k)


def test_bench_sharded_line_carries_roofline_and_ranks_seen():
    """The N > 1 form of bench.py (forced here with the one visible GPU: RCCL process group of one rank, window-sharded
    tracker, pipelined windows): its JSON line must carry what the N = 1 line carries -- `roofline`, `kernels` -- plus
    `ranks_seen`, the sum over ranks of 1 by all-reduce, which the driver can hold against --gpus."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, str(repo / "bench.py"), "--gpus", "1", "--force-sharded", "--steps", "4", "--warmup", "1",
                          "--no-cpu-baseline", "--no-parity", "--no-alt-arith"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["steps"] == 4
    assert "sharded" in d["config"]["parallelism"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and 0.05 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert "lookup_convc1_fused" in d["kernels"] and "conv_gemm" in d["kernels"]
    assert d["pairs_per_frame"]["timed_min"] == 7


@pytest.mark.timeout(900)
def test_bench_two_real_ranks_on_one_gpu():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), both ranks on the one GPU of the
    test box over gloo: the N > 1 path of the bench -- barrier + synchronize fences, MAX over ranks of the timed region, window
    sharding with prefetch and deferred windows, the rank-0 profile pass beside waiting peers, ONE JSON line from rank 0 -- runs with
    two real processes.  The rate of two ranks time-slicing one GPU means nothing and the line says so."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MFT_DIST_BACKEND="gloo", MFT_DIST_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29655", str(REPO / "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=850)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                    # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["steps"] == 6 and d["warmup"] == 2
    assert d["scaling"] == "strong" and d["pairs_per_frame"]["timed_min"] == 7 and d["value"] > 0
    assert "x2" in d["config"]["parallelism"] and "test_only" in d["config"]
    assert d["roofline"]["bound"] == "mfma" and "conv_gemm" in d["kernels"]
