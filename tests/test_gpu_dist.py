"""Delta-sharded tracking over RCCL on the GPUs that are visible (1 on the test
box: the sharded code path -- chain, all-gather, select -- is forced and must
equal the fused single-GPU path bit for bit; with more GPUs visible the same
test runs one rank per GPU)."""
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
def test_delta_sharding_rccl_bitwise(tmp_path):
    n = min(torch.cuda.device_count(), 4)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29621", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", "29621", str(REPO / "tests" / "gpu_dist_worker.py"),
           str(tmp_path)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=550)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    single = np.load(tmp_path / "single.npz")
    for r in range(n):
        got = np.load(tmp_path / f"rank{r}.npz")
        for k in single.files:
            assert np.array_equal(got[k], single[k]), (r, k)
