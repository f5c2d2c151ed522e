#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (run on the GPU box) into small summaries:

    python tools/pmc_summary.py <dir with pmc_FETCH_SIZE/, pmc_WRITE_SIZE/, pmc_MFMA/, trace/> <out dir>

Steady-state launches only = the upper half of each kernel's dispatches by grid
size/duration (7 pairs per frame); values are per launch.
"""
import collections
import csv
import sys
from pathlib import Path

src, out = Path(sys.argv[1]), Path(sys.argv[2])
out.mkdir(parents=True, exist_ok=True)


def short(name):
    return name.split("(")[0].replace("void ", "")


def per_kernel(path, counters):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            if "mftx" in r["Kernel_Name"] and r["Counter_Name"] in counters:
                agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def top_half_mean(v):
    v = sorted(v)
    top = v[len(v) // 2:]
    return sum(top) / len(top)


# ---- HBM traffic
data = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = src / f"pmc_{c}" / "bench_counter_collection.csv"
    if p.exists():
        for k, d in per_kernel(p, {c}).items():
            data.setdefault(k, {})[c] = (len(d[c]), top_half_mean(d[c]) * 1024 / 1e6)
import hashlib
lib = Path(__file__).resolve().parents[1] / "mft_amd" / "libmftx.so"
build = hashlib.sha256(lib.read_bytes()).hexdigest()[:16] if lib.exists() else "unknown"
with open(out / "pmc_hbm_traffic.csv", "w") as f:
    f.write(f"# build: {build}  (sha256 of mft_amd/libmftx.so, first 16 hex digits: bench.py refuses this file for another build)\n")
    f.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) on\n"
            "#   MFTX_SPLIT_STREAMS=1 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-parity --no-host-io --sync-encode\n"
            "#   (512x512, 7 pairs/frame after the 32-frame pre-roll; one batch on one stream so that kernels do not overlap)\n"
            "# per launch, steady-state launches (upper half of each kernel's dispatches); counter x 1024 B.\n"
            "# gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reads 1/2 of the bytes of wide (16 B/lane)\n"
            "# coalesced streams -> column fetch_x2_MB for the kernels that read that way (conv_gemm LDS-DMA, tile_conv's 16-byte loads).\n"
            "kernel,launches,fetch_MB,fetch_x2_MB,write_MB\n")
    for k, v in sorted(data.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", (0, 0))[1]):
        n, fe = v.get("FETCH_SIZE", (0, 0.0))
        _, wr = v.get("WRITE_SIZE", (0, 0.0))
        f.write(f"{k},{n},{fe:.1f},{fe * 2 if ('conv_gemm' in k or 'tile_conv' in k or 'gru_half' in k or 'ou_head' in k) else fe:.1f},{wr:.1f}\n")

# ---- MFMA utilisation
p = src / "pmc_MFMA" / "bench_counter_collection.csv"
if p.exists():
    agg = per_kernel(p, {"SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"})
    with open(out / "pmc_mfma_util.csv", "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES\n"
                "# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), steady-state launches\n"
                "kernel,launches,mfma_busy_Mcycles,gui_active_Mcycles_per_xcd,mfma_util\n")
        for k, d in sorted(agg.items()):
            if not d["SQ_VALU_MFMA_BUSY_CYCLES"] or max(d["SQ_VALU_MFMA_BUSY_CYCLES"]) == 0:
                continue
            busy = top_half_mean(d["SQ_VALU_MFMA_BUSY_CYCLES"])
            gui = top_half_mean(d["GRBM_GUI_ACTIVE"]) / 8
            f.write(f"{k},{len(d['GRBM_GUI_ACTIVE'])},{busy / 1e6:.1f},{gui / 1e6:.3f},{busy / (1024 * gui):.3f}\n")

# ---- kernel stats (top 40)
p = src / "trace" / "bench_kernel_stats.csv"
if p.exists():
    rows = list(csv.reader(open(p)))
    with open(out / "kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        for r in rows[:41]:
            r = list(r)
            r[0] = r[0][:120]
            w.writerow(r)
# ---- steady-state kernel table from the raw trace: the tracker issues the same number of launches per kernel for every
# frame (only the pair count, i.e. the grid, grows over the 32 pre-roll frames), so the LAST third of every kernel's
# dispatches (by start time) belongs to the 7-pair frames bench.py times -- comparable with its HIP-event averages
p = src / "trace" / "bench_kernel_trace.csv"
if p.exists():
    per = collections.defaultdict(list)
    with open(p) as f:
        for r in csv.DictReader(f):
            if "mftx" in r["Kernel_Name"]:
                per[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    rows = []
    for k, v in per.items():
        v.sort()
        tail = [d for _, d in v[len(v) - max(1, len(v) // 3):]]
        rows.append((sum(tail), k, len(v), len(tail), sum(tail) / len(tail) / 1e3))
    tot = sum(r[0] for r in rows)
    with open(out / "kernel_stats_steady.csv", "w") as f:
        f.write("# steady-state (7 pairs per frame) launches only: last third of each kernel's dispatches in the rocprofv3\n"
                "# --kernel-trace of tools/gpu_profile.sh's bench command (one batch on one stream, encoders on the main stream)\n"
                "kernel,dispatches_total,dispatches_steady,avg_us_steady,share_of_steady_gpu_time\n")
        for t, k, n, m, avg in sorted(rows, reverse=True):
            f.write(f'"{k}",{n},{m},{avg:.1f},{t / tot:.4f}\n')
        gemm = [(t, m) for t, k, n, m, avg in rows if ("conv_gemm" in k and ", 4, 32" not in k) or "tile_conv_kernel" in k or "tile_conv2p_kernel" in k or "gru_half_kernel" in k or "ou_head_kernel" in k]
        if gemm:
            f.write(f"# all conv-GEMM launches (ring-buffered and tile-resident) except the volume GEMM: {sum(t for t, _ in gemm) / sum(m for _, m in gemm) / 1e3:.1f} us "
                    f"average -- the figure bench.py reports as kernels.conv_gemm.avg_us (HIP events, + ~1.5 us of bracket)\n")
print("wrote", sorted(x.name for x in out.iterdir()))
