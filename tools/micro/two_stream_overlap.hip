// Do two streams' kernels run at the same time when each kernel is ONE round of workgroups that own a CU through their LDS -- the shape
// of the tracker's loop kernels (224 workgroups x 512 threads x 141 KB of LDS on 256 CUs)?  Each kernel spins for a fixed number of
// cycles.  N kernels on each of two streams against 2 N kernels on one stream, for several grid sizes and LDS footprints: if the
// 32 idle CUs (and the tails) were used by the other stream's kernel, two streams would be faster.
//   hipcc --offload-arch=gfx950 -O3 two_stream_overlap.hip -o two_stream_overlap && ./two_stream_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(512) void spin(unsigned long long cycles, float *out) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    unsigned long long t0, t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)); } while (t - t0 < cycles);
    if (lds[(threadIdx.x + 1) & 511] < -1.f) out[0] = 1.f;
}

static double run(int streams, int per_stream, int grid, int lds, unsigned long long cycles, float *d) {
    hipStream_t s[2];
    for (int i = 0; i < 2; ++i) hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
    hipFuncSetAttribute(reinterpret_cast<const void *>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceSynchronize();
    hipEventRecord(e0, s[0]);
    hipStreamWaitEvent(s[1], e0, 0);
    for (int k = 0; k < per_stream; ++k)
        for (int i = 0; i < streams; ++i) hipLaunchKernelGGL(spin, dim3(grid), dim3(512), lds, s[i], cycles, d);
    hipEvent_t j; hipEventCreate(&j); hipEventRecord(j, s[1]); hipStreamWaitEvent(s[0], j, 0);
    hipEventRecord(e1, s[0]);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    for (int i = 0; i < 2; ++i) hipStreamDestroy(s[i]);
    return ms;
}

int main() {
    float *d; hipMalloc(&d, 4);
    const unsigned long long cycles = 100000;   // s_memtime counts shader cycles: ~50 us at ~2 GHz
    printf("# N = 50 kernels of ~50 us per stream; ms for [one stream x 2 N] vs [two streams x N]\n");
    for (int lds : {141 * 1024, 70 * 1024, 16 * 1024})
        for (int grid : {224, 128, 64}) {
            const double one = run(1, 100, grid, lds, cycles, d), two = run(2, 50, grid, lds, cycles, d);
            printf("grid %3d, LDS %3d KB: one stream %7.2f ms, two streams %7.2f ms (%.2f x)\n", grid, lds / 1024, one, two, one / two);
        }
    return 0;
}
