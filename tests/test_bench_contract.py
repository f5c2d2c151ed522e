"""bench.py's host-side helpers (no GPU): the committed PMC summary parses into the roofline's
`traffic`, the CLI keeps the driver's flags, the oracle thread cap holds."""
import json
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))


def test_profiled_traffic_reads_the_committed_pmc_summary():
    import bench
    traffic, source = bench.profiled_traffic()
    assert source.endswith("_pmc_hbm_traffic.csv") and (REPO / "profiles" / source).exists()
    # conv GEMM launches move tens of MB each (fetch x2-corrected + write), never the volume's 0.9 GB
    assert 2e7 < traffic < 3e8
    text = (REPO / "profiles" / source).read_text()
    assert "conv_gemm_kernel" in text and "fetch_x2_MB" in text


def test_cli_flags_of_the_driver_contract():
    out = subprocess.run([sys.executable, str(REPO / "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout


def test_oracle_thread_cap(monkeypatch):
    import bench
    monkeypatch.setenv("MFT_ORACLE_THREADS", "4")
    assert bench.oracle_threads() <= 4
    monkeypatch.delenv("MFT_ORACLE_THREADS")
    assert 1 <= bench.oracle_threads() <= 16


def test_committed_bench_line_has_the_contract_keys():
    lines = sorted((REPO / "profiles").glob("r[0-9]*_bench.json"))
    d = json.loads(lines[-1].read_text())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["config"]["workload"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["sample"]
