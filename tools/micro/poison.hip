// Poison the state a kernel must not depend on: every CU's LDS and the vector registers of every SIMD are filled with a NaN pattern
// by workgroups that occupy a whole CU each.  A kernel that reads LDS or registers it never wrote changes its result when this runs
// in front of it (tools/race_kernels.py --poison).   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o poison.so poison.hip
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(1024) void poison_kernel(unsigned pattern, unsigned *sink) {
    extern __shared__ unsigned lds[];
    const int n = 160 * 1024 / 4;
    for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = pattern;
    __syncthreads();
    // 100+ live registers per lane holding the pattern (1024 threads = 4 waves per SIMD x 128 VGPRs = the whole file)
    unsigned r[100];
#pragma unroll
    for (int i = 0; i < 100; ++i) r[i] = pattern + (lds[(threadIdx.x + i) % n] & 0u);
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 100; ++i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(acc) : "v"(r[i]));
    if (acc == 0x12345u) sink[0] = acc;
}

extern "C" int poison_launch(unsigned pattern, unsigned *sink, void *stream) {
    static bool set = false;
    if (!set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
        set = true;
    }
    hipLaunchKernelGGL(poison_kernel, dim3(1024), dim3(1024), 160 * 1024, static_cast<hipStream_t>(stream), pattern, sink);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
