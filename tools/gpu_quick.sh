#!/bin/bash
# quick GPU loop: the tests named by $1 (a -k expression), the GRU-pass trace, one bench line summarised
K="${1:-gru}"
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_random.py -x -q -m gpu -k "$K" 2>&1 | tail -3
if [ -f build_tune/libmftx_tune.so ] && [ -z "$NO_TRACE" ]; then
  (for a in "7" "7 1"; do MFTX_LIB=$PWD/build_tune/libmftx_tune.so timeout 200 python tools/gru_trace.py $a 2>&1 | grep -v amdgpu.ids; done) > gpurun_out/gru_trace_new.txt; cat gpurun_out/gru_trace_new.txt
fi
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_q.json 2>gpurun_out/bench_q.err
python - <<'P'
import json
d = json.load(open("gpurun_out/bench_q.json"))
k = d["kernels"]
print("fps %.1f ms %.3f frac %.3f epe %.2e" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["parity"]["flow_epe_px"]))
print({n: round(v["avg_us"], 1) for n, v in k.items()})
P
