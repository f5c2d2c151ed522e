mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -4 > gpurun_out/peel_test.txt
for i in 1 2; do
MFTX_LIB=$PWD/mft_amd/csrc/abl/libmftx_old.so timeout 600 python bench.py --no-alt-arith --no-cpu-baseline --no-host-io 2>/dev/null | tail -1 > gpurun_out/peel_a$i.json
timeout 600 python bench.py --no-alt-arith --no-cpu-baseline --no-host-io 2>/dev/null | tail -1 > gpurun_out/peel_b$i.json
done
MFTX_LIB=$PWD/mft_amd/csrc/abl/libmftx_old.so timeout 300 python tools/bench_conv.py --arith 1 --a-split --out-split 2>&1 | grep -E "us " > gpurun_out/peel_conv_old.txt
timeout 300 python tools/bench_conv.py --arith 1 --a-split --out-split 2>&1 | grep -E "us " > gpurun_out/peel_conv_new.txt
