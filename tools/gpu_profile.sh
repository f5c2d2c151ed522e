#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/gpu_profile.sh <tag>          -> gpurun_out/<tag>/...  (copy what is to be judged into profiles/)
# PMC passes run with --kernel-trace only (no other trace domains), one counter set per pass.
set -u
TAG=${1:-r2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --steps 20 --warmup 5"
# per-kernel passes: one batch, one stream, encoders on the main stream -> every kernel runs alone, like bench.py's own
# HIP-event pass (the timed region of the default run overlaps two half-batches and the next frame's encoders)
QUIET="--no-cpu-baseline --no-profile --no-parity --no-host-io --sync-encode --no-alt-arith"

$BENCH > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
$BENCH --arith fp32 --no-cpu-baseline --no-host-io --no-alt-arith > $OUT/bench_fp32_arith.json 2> $OUT/bench_fp32.err; echo "bench fp32 rc=$?"
for a in 0 1; do echo "== arithmetic $a (0 fp32 MFMA, 1 split fp16): error against an fp64 convolution"; timeout 200 python tools/conv_arith_error.py $a 2>&1 | grep -v amdgpu.ids; done > $OUT/conv_arith_error.txt
(timeout 100 tools/micro/mfma_shadow; timeout 100 tools/micro/lds_fill) > $OUT/micro_mfma_shadow_lds_fill.txt 2>&1
for t in bench_lookup bench_small bench_corr; do timeout 120 python tools/$t.py 2>&1 | grep -v amdgpu.ids; done > $OUT/micro.txt
timeout 300 python tools/bench_pairs.py > $OUT/bench_pairs.txt 2>/dev/null
(echo "== split fp16 arithmetic"; timeout 300 python tools/bench_conv.py --P 7 --arith 1; echo "== fp32 MFMA"; timeout 300 python tools/bench_conv.py --P 7 --arith 0; echo "== encoder layers, split"; timeout 100 python tools/bench_conv.py --enc --arith 1) 2>/dev/null | grep -v amdgpu.ids > $OUT/bench_conv_P7.txt

MFTX_SPLIT_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof/trace -o bench -- $BENCH $QUIET > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  MFTX_SPLIT_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/prof/pmc_$c -o bench -- python bench.py --steps 3 --warmup 2 $QUIET > /dev/null 2>&1
done
MFTX_SPLIT_STREAMS=1 timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/prof/pmc_MFMA -o bench -- python bench.py --steps 3 --warmup 2 $QUIET > /dev/null 2>&1
# rocprofv3 nests its output under <hostname>/: flatten
for d in trace pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_MFMA; do find $OUT/prof/$d -name '*.csv' -exec mv {} $OUT/prof/$d/ \; 2>/dev/null; done
python tools/pmc_summary.py $OUT/prof $OUT
# lookup: memory-side request counts
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $set | md5sum | cut -c1-6)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/prof/lk_$n -o lk -- python tools/bench_lookup.py > /dev/null 2>&1
done
python tools/pmc_kernel_table.py $OUT/prof corr_lookup > $OUT/pmc_lookup_requests_raw.txt 2>&1
rm -rf $OUT/prof/*/*/   # the per-host raw directories (large)
du -sh $OUT
