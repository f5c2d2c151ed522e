#!/bin/bash
# Per-kernel times of one bench.py run (rocprofv3 --kernel-trace --stats; every kernel alone: one lane, encoders on the main stream, no graphs)
#   bash tools/kstats.sh <out-dir> [extra bench.py flags]   -> <out-dir>/kernel_stats.csv + a 40-line summary on stdout
set -u
OUT=$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
D=$(mktemp -d /tmp/kstats.XXXX)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile \
    --no-parity --no-host-io --sync-encode --no-alt-arith --no-graphs --frames-in-flight 1 "$@" > /dev/null 2>&1 < /dev/null
f=$(find $D -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -z "$f" ]; then echo "no kernel_stats.csv under $D"; ls -R $D | head; exit 1; fi
cp "$f" $OUT/kernel_stats.csv
python - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    name = re.sub(r"\(.*", "", r["Name"])[:90]
    print(f'{name:90s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"]) / 1e3:8.1f} us total {float(r["TotalDurationNs"]) / 1e6:8.2f} ms {float(r["Percentage"]):5.1f} %')
PY
