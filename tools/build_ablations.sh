#!/bin/bash
# Tuning builds of libmftx.so with parts of the split-arithmetic K loop removed (-DMFTX_SABL=bits, see conv_gemm.hip):
#   (T: per-phase stamps inside the K loop, perturbing; P: tile phases only -- prologue / K loop / epilogue)
#   bash tools/build_ablations.sh 1 3 7 T P ...   ->  gpurun_out/../mft_amd/csrc/abl/libmftx_<bits>.so ; select with MFTX_LIB=<path>
set -e
cd "$(dirname "$0")/../mft_amd/csrc"
mkdir -p abl
for b in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DMFTX_CONV_SINGLE_TU -DMFTX_SABL=${b/[TP]/0} $( [ "$b" = T ] && echo -DMFTX_TIMING=1 ) $( [ "$b" = P ] && echo -DMFTX_TIMING=2 ) -c conv_gemm.hip -o abl/conv_gemm_$b.o &
done
wait
for b in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o abl/libmftx_$b.so api.o abl/conv_gemm_$b.o conv_small.o corr.o corr_ondemand.o lookup_convc1.o flow_branch.o tile_conv.o volume_tile.o upsample.o chain.o raft_engine.o encoder.o codec.o
done
ls -la abl/*.so
