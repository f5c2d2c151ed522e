// How fast do 224 workgroups of 8 waves write a [28672][256] fp32 map (29 MB), by store pattern?
//   A: as the conv GEMM epilogue does -- a wave owns a 64 x 64 sub-tile, one float4 store instruction covers 8 rows x 128 B
//      (rows 1 KiB apart);
//   B: a wave owns whole rows -- one instruction covers one row, 1 KiB contiguous;
//   C: as A with 32 x 64 waves (128 x 128 tile, ld = 128): 8 rows x 128 B, rows 512 B apart.
// Eight maps in rotation (233 MB: beyond L2, about the MALL).  hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(512) void k(float *out, float v) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float *tile = out + (size_t)blockIdx.x * 128 * 256;
    const f32x4 val = {v, v + 1, v + 2, v + 3};
    if (PAT == 0) {
        const int wm = w >> 2, wn = w & 3;
#pragma unroll
        for (int ij = 0; ij < 4; ++ij)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = wm * 64 + (ij & 1) * 32 + t * 8 + (lane >> 3), col = wn * 64 + (ij >> 1) * 32 + (lane & 7) * 4;
                *reinterpret_cast<f32x4 *>(tile + row * 256 + col) = val;
            }
    } else if (PAT == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) *reinterpret_cast<f32x4 *>(tile + (w * 16 + r) * 256 + lane * 4) = val;
    } else {
        // two 128 x 128 tiles side by side in memory terms: [256 rows][128]
        const int wm = w >> 1, wn = w & 1;
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int row = half * 128 + wm * 32 + t * 8 + (lane >> 3), col = wn * 64 + j * 32 + (lane & 7) * 4;
                    *reinterpret_cast<f32x4 *>(tile + row * 128 + col) = val;
                }
    }
}

template <int PAT>
static void run(const char *name, float *buf) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t map = (size_t)28672 * 256;
    for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(k<PAT>, dim3(224), dim3(512), 0, 0, buf + (i % 8) * map, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 64;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<PAT>, dim3(224), dim3(512), 0, 0, buf + (i % 8) * map, (float)i);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-60s %6.2f us per 29.4 MB map = %5.2f TB/s\n", name, ms * 1e3 / reps, map * 4.0 * reps / (ms * 1e-3) / 1e12);
}

int main() {
    float *buf; hipMalloc(&buf, (size_t)8 * 28672 * 256 * 4);
    run<0>("A: 64 x 64 wave sub-tiles, 8 rows x 128 B per instruction", buf);
    run<1>("B: whole rows, 1 KiB per instruction", buf);
    run<2>("C: 32 x 64 wave sub-tiles of 128-wide maps", buf);
    run<0>("A again", buf);
    run<1>("B again", buf);
    return 0;
}
