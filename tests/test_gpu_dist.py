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
        enc = 0
        for r in range(n):
            got = np.load(tmp_path / f"rank{r}_{mode}.npz")
            enc += int(got["_encoded"])
            for k in single.files:
                assert np.array_equal(got[k], single[k]), (mode, r, k)
        if mode in ("L6", "L6p", "L6d") and n <= 6:
            # every exchanged frame is encoded exactly once across the ranks (the ragged last window of one
            # frame is shorter than the world size when n > 1: every rank encodes it itself)
            assert enc == (13 if n == 1 else 12), enc


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
