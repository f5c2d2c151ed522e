#!/bin/bash
# tracker-level determinism under GPU contention: two load generators beside the probes
for i in 0 1; do python tools/race_kernels.py --load-seconds ${LOAD_S:-130} --tag load$i > gpurun_out/load$i.txt 2>&1 & done
sleep 10
python tools/race_chain2.py 150 2>&1 | grep -v amdgpu.ids | head -4
python tools/race_lanes.py --reps 12 --tag A 2>&1 | grep -v amdgpu.ids | cut -c1-200
python tools/race_lanes.py --reps 12 --sync --numpy --tag B 2>&1 | grep -v amdgpu.ids | cut -c1-200
wait
