#!/bin/bash
# Issue / stall counters of one conv layer (run through gpurun):  bash tools/pmc_conv_stalls.sh <tile> <arith> "<layer substring>"
set -u
TILE=${1:-6}; AR=${2:-1}; LAYER=${3:-gru zr 1x5}
OUT=gpurun_out/stalls_t${TILE}_a${AR}
mkdir -p $OUT; export TMPDIR=/tmp
CMD="python tools/bench_conv.py --P 7 --arith $AR --a-split --out-split --reps 5 --only"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  MFTX_CONV_TILE=$TILE timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o c -- $CMD "$LAYER" > /dev/null 2>&1
done
python tools/pmc_kernel_table.py $OUT conv_gemm > $OUT/table.txt 2>&1
rm -rf $OUT/p*/
cat $OUT/table.txt
