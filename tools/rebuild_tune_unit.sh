#!/bin/bash
# Recompile ONE translation unit of the tuning build (tools/build_tuning.sh made the rest) and relink: bash tools/rebuild_tune_unit.sh lookup_convc1 "-DMFTX_LF_TRACE"
set -e
cd "$(dirname "$0")/.."
u=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DMFTX_TUNING -DMFTX_EXPERIMENTAL_TILES "$@" -c mft_amd/csrc/$u.hip -o build_tune/obj/$u.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_tune/libmftx_tune.so build_tune/obj/*.o
