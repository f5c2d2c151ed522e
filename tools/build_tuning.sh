#!/bin/bash
# Tuning build of libmftx (build_tune/libmftx_tune.so): the product sources with -DMFTX_TUNING, i.e. with the
# environment switches of tools/ (MFTX_CONV_TILE, MFTX_LOOKUP_ABLATE, MFTX_LF_ABLATE, ...) compiled in, plus the
# measurement-only tile shapes.  Use with MFTX_LIB=build_tune/libmftx_tune.so.  Extra flags: $1.
set -e
cd "$(dirname "$0")/.."
SRC=mft_amd/csrc; OBJ=build_tune/obj; mkdir -p $OBJ
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DMFTX_TUNING -DMFTX_EXPERIMENTAL_TILES $1"
jobs=()
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"     # as csrc/Makefile: no packed-fp32 code in the units with EXEC-masked gathers
for f in conv_small lookup_convc1 flow_branch tile_conv volume_tile conv_gemm; do jobs+=("$f.hip|$f.o|"); done
for f in corr corr_ondemand upsample encoder raft_engine; do jobs+=("$f.hip|$f.o|$NOPK"); done
for k in 0 1 2 3; do jobs+=("conv_gemm.hip|conv_gemm_p$k.o|-DMFTX_CONV_PART=$k"); done
jobs+=("chain.hip|chain.o|-ffp-contract=off -fno-slp-vectorize $NOPK" "codec.hip|codec.o|-ffp-contract=off $NOPK" "api.cpp|api.o|-x hip")
printf '%s\n' "${jobs[@]}" | xargs -P 8 -I{} bash -c 'IFS="|" read s o x <<< "{}"; /opt/rocm/bin/hipcc '"$FLAGS"' $x -c '"$SRC"'/$s -o '"$OBJ"'/$o'
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_tune/libmftx_tune.so $OBJ/*.o
ls -la build_tune/libmftx_tune.so
