"""bench.py's host-side helpers (no GPU): the committed PMC summary parses into the roofline's
`traffic`, the CLI keeps the driver's flags, the oracle thread cap holds."""
import json
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))


def test_profiled_traffic_is_tied_to_the_build(tmp_path):
    """`roofline.traffic` comes from a rocprofv3 counter pass (tools/gpu_profile.sh -> profiles/*_pmc_hbm_traffic.csv); the file
    names the build it was measured on and bench.py refuses it for any other (no stale counter files)."""
    import bench
    mine = bench.build_hash()
    assert mine and len(mine) == 16
    body = ("# rocprofv3 ...\nkernel,launches,fetch_MB,fetch_x2_MB,write_MB\n"
            '"mftx::conv_gemm_kernel<128, 192, 4, 2, 1, 32, 2, 3, 0>",24,20.0,40.0,22.0\n'
            '"mftx::gru_half_kernel<2, 64, 1, 5>",24,30.0,60.0,18.0\n'
            '"mftx::ou_head_kernel<128, 5>",24,40.0,80.0,10.0\n'
            '"mftx::volume_tile_kernel",2,200.0,200.0,600.0\n')
    (tmp_path / "r9z_pmc_hbm_traffic.csv").write_text(f"# build: {mine}  (sha256 ...)\n" + body)
    traffic, source = bench.profiled_traffic(tmp_path)
    assert abs(traffic - (62e6 + 78e6 + 90e6) / 3) < 1 and source.startswith("r9z_pmc_hbm_traffic.csv") and mine in source
    (tmp_path / "r9z_pmc_hbm_traffic.csv").write_text("# build: 0123456789abcdef\n" + body)
    traffic, source = bench.profiled_traffic(tmp_path)
    assert traffic is None and "refused" in source
    (tmp_path / "r9z_pmc_hbm_traffic.csv").write_text(body)              # a file from before the rule: no build line
    assert bench.profiled_traffic(tmp_path)[0] is None
    # the committed file parses (whether or not it belongs to the library built here)
    traffic, source = bench.profiled_traffic()
    assert source and (traffic is None or 2e7 < traffic < 3e8)


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
