#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/gpu_profile.sh <tag>          -> gpurun_out/<tag>/...  (copy what is to be judged into profiles/)
# PMC passes run with --kernel-trace only (no other trace domains), one counter set per pass.
set -u
TAG=${1:-r3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --steps 20 --warmup 5"
# per-kernel passes: one batch, one stream, encoders on the main stream -> every kernel runs alone, like bench.py's own
# HIP-event pass (the timed region of the default run overlaps two half-batches and the next frame's encoders)
QUIET="--no-cpu-baseline --no-profile --no-parity --no-host-io --sync-encode --no-alt-arith --no-graphs --frames-in-flight 1"

$BENCH > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
# A/B of the round's structural changes, same box, same run (results do not depend on any of them beyond fp32 rounding)
$BENCH --no-cpu-baseline --no-parity --no-alt-arith --no-graphs > $OUT/bench_no_graphs.json 2>/dev/null; echo "bench no-graphs rc=$?"
$BENCH --no-cpu-baseline --no-parity --no-alt-arith --no-fused-lookup > $OUT/bench_no_fused_lookup.json 2>/dev/null; echo "bench no-fused-lookup rc=$?"
(for o in "" "tile_conv2p=0" "fuse_gru=0" "fuse_ou=0" "gather=0" "fuse_head=2" "fuse_head=0" "tile_volume=0" "tile_conv=0" "fuse_flow=0" "fuse_lookup=0" "fuse_head=0 tile_conv=0 fuse_flow=0" "gather=0 tile_volume=0 fuse_head=0 tile_conv=0 fuse_flow=0 fuse_lookup=0 fuse_gru=0 fuse_ou=0 graph=0"; do
   args=""; for kv in $o; do args="$args --engine-opt $kv"; done
   for rep in 1 2; do $BENCH --no-cpu-baseline --no-parity --no-alt-arith --no-host-io $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('engine options [$o]:', round(d['value'],1), 'frames/s, host', round(d['host_enqueue_ms_per_step'],2), 'ms per frame, conv GEMM', round(d['roofline']['frac'],3), 'of the fp16 MFMA peak')"; done
 done) > $OUT/engine_options_ab.txt
# the fused lookup's 16-byte-gather variant (csrc/lookup_convc1_wide.hip -> mft_amd/libmftx_lfwide.so) against the default, alternating (round 6)
if [ -f mft_amd/libmftx_lfwide.so ]; then
  (for rep in 1 2 3; do for lib in "" "MFTX_LIB=$PWD/mft_amd/libmftx_lfwide.so"; do env $lib $BENCH --no-cpu-baseline --no-parity --no-alt-arith --no-host-io 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels']['lookup_convc1_fused']; print('${lib:+16-byte gather (lfwide)}' or 'dword gather (default)  ', round(d['value'],1), 'frames/s; lookup_convc1_fused', round(k['avg_us'],1), 'us, frac', round(k['frac'],3))"; done; done) > $OUT/lookup_gather_ab.txt
fi
# frames in flight (flow_config.frames_in_flight): 1 / 2 / 3 lanes, same box, twice
(for rep in 1 2; do for f in 1 2 3; do $BENCH --no-cpu-baseline --no-parity --no-alt-arith --no-profile --frames-in-flight $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('frames in flight $f:', round(d['value'],1), 'frames/s,', round(d['ms_per_step'],3), 'ms per frame, with PCIe', round(d['host_io_fps'],1))"; done; done
 for f in 1 2; do python bench.py --steps 20 --warmup 5 --height 256 --width 256 --no-cpu-baseline --no-parity --no-alt-arith --no-profile --no-host-io --frames-in-flight $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('256 x 256, frames in flight $f:', round(d['value'],1), 'frames/s')"; done
 for f in 1 2; do python bench.py --steps 6 --warmup 2 --height 1080 --width 1920 --no-cpu-baseline --no-parity --no-alt-arith --no-profile --no-host-io --frames-in-flight $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('1080p, frames in flight $f:', round(d['value'],2), 'frames/s')"; done) > $OUT/frames_in_flight_ab.txt
python bench.py --steps 20 --warmup 5 --height 256 --width 256 --no-cpu-baseline --no-parity --no-alt-arith > $OUT/bench_256.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --height 1080 --width 1920 --no-cpu-baseline --no-parity --no-alt-arith --no-host-io > $OUT/bench_1080p.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --force-sharded --no-cpu-baseline --no-parity --no-alt-arith > $OUT/bench_forced_sharded_x1.json 2>/dev/null
(for g in 2 4 8; do python bench.py --steps 20 --warmup 5 --force-sharded --emulate-world $g --no-cpu-baseline --no-parity --no-alt-arith --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('emulated world $g (driver flags):', round(d['value'],1), 'frames/s')"; done; python bench.py --steps 80 --warmup 5 --force-sharded --emulate-world 8 --no-cpu-baseline --no-parity --no-alt-arith --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('emulated world 8 (80 steps):', round(d['value'],1), 'frames/s')"
 for g in 2 4 8; do python bench.py --steps 40 --warmup 8 --force-sharded --emulate-world $g --window 1 --no-cpu-baseline --no-parity --no-alt-arith --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('per-frame mode (--window 1), emulated world $g:', round(d['value'],1), 'frames/s')"; done) > $OUT/emulated_world.txt
IO_ALL=1 timeout 300 python tools/io_paths3.py 2>&1 | grep -v amdgpu.ids > $OUT/io_paths.txt; IO_ALL=1 IO_THREADS=1 timeout 300 python tools/io_paths3.py 2>&1 | grep -v amdgpu.ids >> $OUT/io_paths.txt
timeout 100 python tools/io_kernel_copy.py 2>&1 | grep -v amdgpu.ids >> $OUT/io_paths.txt
timeout 300 python tools/lf_stress.py 2>&1 | grep -v amdgpu.ids > $OUT/lf_stress.txt
(for s in "7 64 64" "1 64 64" "7 136 240"; do timeout 200 python tools/bench_lookup_fused.py $s 2>&1 | tail -1; done) > $OUT/lookup_fused_micro.txt
(for s in "7 64 64" "1 64 64" "7 135 240"; do timeout 200 python tools/bench_flow_branch.py $s 2>&1 | tail -1; done) > $OUT/flow_branch_micro.txt
timeout 300 python tools/bench_tile_conv.py 2>&1 | grep -v amdgpu.ids > $OUT/tile_conv_micro.txt
(timeout 200 python tools/bench_volume.py; timeout 200 python tools/bench_volume.py 2 135 240) 2>&1 | grep -v amdgpu.ids > $OUT/volume_micro.txt
(timeout 200 tools/micro/mfma_power; timeout 200 tools/micro/mfma_energy) > $OUT/mfma_power_energy.txt 2>&1
if [ -f build_tune/libmftx_tune.so ]; then
  MFTX_LIB=$PWD/build_tune/libmftx_tune.so timeout 400 python tools/clock_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-320 > $OUT/clock.txt
  (echo "# tools/gru_trace.py: s_memtime stamps of workgroup 0 of gru_half_kernel (kilo-cycles), 30th launch of a back-to-back loop"; for a in "7" "7 1" "1"; do MFTX_LIB=$PWD/build_tune/libmftx_tune.so timeout 200 python tools/gru_trace.py $a 2>&1 | grep -v amdgpu.ids; done) > $OUT/gru_trace.txt
  (echo "# tools/lf_trace.py on a -DMFTX_TUNING -DMFTX_LF_TRACE build (tools/build_tuning.sh): s_memtime stamps of workgroup 0, P = 7, 64 x 64"
   MFTX_LIB=$PWD/build_tune/libmftx_tune.so timeout 200 python tools/lf_trace.py 2>&1 | grep -v amdgpu.ids
   for a in 1 2 8 10 32 64 128 512 1024; do echo "== MFTX_LF_ABLATE=$a"; MFTX_LIB=$PWD/build_tune/libmftx_tune.so MFTX_LF_ABLATE=$a timeout 200 python tools/lf_trace.py 2>&1 | tail -4; done) > $OUT/lf_trace.txt
  (echo "# tools/fb_trace.py: s_memtime stamps of workgroup 0 of the fused flow-branch kernel, P = 7, 64 x 64 (kilo-cycles)"
   MFTX_LIB=$PWD/build_tune/libmftx_tune.so timeout 200 python tools/fb_trace.py 2>&1 | grep -v amdgpu.ids) > $OUT/fb_trace.txt
  (echo "# tools/tc_trace.py: s_memtime stamps of workgroup 0 of the tile-resident conv kernel (flow head first layer, 3 x 3, 128 -> 256) with P pairs of 64 x 64 cells"
   echo "# = 32 P workgroups on the 256 CUs: the SAME cycles per workgroup take longer the more CUs are busy -- the clock follows the chip's power"
   for P in 1 2 4 7; do echo "== P = $P"; MFTX_LIB=$PWD/build_tune/libmftx_tune.so timeout 200 python tools/tc_trace.py 128 256 3 3 $P 2>&1 | grep -v amdgpu.ids | sed -n "1,2p;9p"; done
   echo "== GRU z | r gates (1 x 5, 256 -> 256), P = 7"; MFTX_LIB=$PWD/build_tune/libmftx_tune.so timeout 200 python tools/tc_trace.py 256 256 1 5 7 2>&1 | grep -v amdgpu.ids | sed -n "1,2p;9p"
   echo "== GRU q gate (1 x 5, 256 -> 128), P = 7"; MFTX_LIB=$PWD/build_tune/libmftx_tune.so timeout 200 python tools/tc_trace.py 256 128 1 5 7 2>&1 | grep -v amdgpu.ids | sed -n "1,2p;9p") > $OUT/tc_trace.txt
fi
$BENCH --arith fp32 --no-cpu-baseline --no-host-io --no-alt-arith > $OUT/bench_fp32_arith.json 2> $OUT/bench_fp32.err; echo "bench fp32 rc=$?"
for a in 0 1; do echo "== arithmetic $a (0 fp32 MFMA, 1 split fp16): error against an fp64 convolution"; timeout 200 python tools/conv_arith_error.py $a 2>&1 | grep -v amdgpu.ids; done > $OUT/conv_arith_error.txt
(timeout 100 tools/micro/mfma_shadow; timeout 100 tools/micro/lds_fill) > $OUT/micro_mfma_shadow_lds_fill.txt 2>&1
for t in bench_lookup bench_small bench_corr; do timeout 120 python tools/$t.py 2>&1 | grep -v amdgpu.ids; done > $OUT/micro.txt
timeout 300 python tools/bench_pairs.py > $OUT/bench_pairs.txt 2>/dev/null
(echo "== split fp16 arithmetic, A and outputs in split form (what the engine runs)"; timeout 300 python tools/bench_conv.py --P 7 --arith 1 --a-split --out-split; echo "== the same on the warp-specialised 128 x 128 tile (11)"; timeout 300 python tools/bench_conv.py --P 7 --arith 1 --a-split --out-split --tile 11; echo "== split fp16 arithmetic, fp32 operands"; timeout 300 python tools/bench_conv.py --P 7 --arith 1; echo "== fp32 MFMA"; timeout 300 python tools/bench_conv.py --P 7 --arith 0; echo "== encoder layers, split"; timeout 100 python tools/bench_conv.py --enc --arith 1) 2>/dev/null | grep -v amdgpu.ids > $OUT/bench_conv_P7.txt

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
# the same memory-side counters, and the texture unit's, for the fused lookup + convc1 kernel
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_READ_sum" "TA_BUSY_avr TA_TA_BUSY_sum TA_BUFFER_LOAD_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS"; do
  n=$(echo $set | md5sum | cut -c1-6)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/prof/lf_$n -o lf -- python tools/bench_lookup_fused.py > /dev/null 2>&1
done
python tools/pmc_kernel_table.py $OUT/prof lookup_convc1 > $OUT/pmc_lookup_fused_raw.txt 2>&1
# ... and where the pyramid is HBM-resident: 7 pairs of 1080p (136 x 240 cells)
mkdir -p $OUT/prof1080
for set in "FETCH_SIZE" "WRITE_SIZE" "TA_BUSY_avr TA_TA_BUSY_sum TA_BUFFER_LOAD_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES"; do
  n=$(echo $set | md5sum | cut -c1-6)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/prof1080/lf_$n -o lf -- python tools/bench_lookup_fused.py 7 136 240 > /dev/null 2>&1
done
python tools/pmc_kernel_table.py $OUT/prof1080 lookup_convc1 > $OUT/pmc_lookup_fused_1080p_raw.txt 2>&1
rm -rf $OUT/prof1080/*/*/
# determinism under contention (round 5): three concurrent processes, every engine option set; kernels under two load generators
(for i in 0 1 2; do python tools/race_probe.py --reps 100 --tag p$i > $OUT/race_probe_$i.txt 2>&1 & done; wait; cat $OUT/race_probe_?.txt | grep distinct; rm -f $OUT/race_probe_?.txt
 for i in 1 2; do python tools/race_kernels.py --load-seconds 35 --tag load$i > $OUT/race_load_$i.txt 2>&1 & done; sleep 8; python tools/race_kernels.py --reps 2000 --tag under-load 2>&1 | grep distinct; wait; cat $OUT/race_load_?.txt | grep "load:"; rm -f $OUT/race_load_?.txt) > $OUT/race_contention.txt 2>&1
# ... and the tracker level (chain + selection, one and two frames in flight) beside two load generators
LOAD_S=130 bash tools/lane_stress.sh > $OUT/race_tracker_contention.txt 2>&1
rm -rf $OUT/prof/*/*/   # the per-host raw directories (large)
du -sh $OUT
