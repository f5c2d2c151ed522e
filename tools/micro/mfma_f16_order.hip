// Does v_mfma_f32_16x16x32_f16 (32 k per instruction) round like two v_mfma_f32_32x32x16_f16 (16 k each) on the same
// data?  (A 16-row tile shape would need it to keep results independent of the tile shape.)  One wave: C = A B^T for
// A, B of 32 rows x KT k (chained accumulation over KT / 32 steps), fp16 values with wide dynamic range so that
// the rounding of the accumulation shows.
//   hipcc --offload-arch=gfx950 -O3 mfma_f16_order.hip -o mfma_f16_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int KT = 256;
__global__ void k(const _Float16 *A, const _Float16 *B, float *c32, float *c16) {   // A, B: [32][KT] row-major (k contiguous)
    const int lane = threadIdx.x;
    {   // 32x32x16 twice: lane = row (l % 32), k half (l / 32) of a 16-wide group
        f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int g = 0; g < KT / 16; ++g) {
            f16x8 a, b;
            for (int e = 0; e < 8; ++e) { a[e] = A[(lane & 31) * KT + 16 * g + 8 * (lane >> 5) + e]; b[e] = B[(lane & 31) * KT + 16 * g + 8 * (lane >> 5) + e]; }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
        for (int r = 0; r < 16; ++r) c32[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[r];
    }
    for (int ti = 0; ti < 2; ++ti)
        for (int tj = 0; tj < 2; ++tj) {   // 16x16x32 once per 16x16 block: lane = row (l % 16), k quarter (l / 16)
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int g = 0; g < KT / 32; ++g) {
                f16x8 a, b;
                for (int e = 0; e < 8; ++e) { a[e] = A[(16 * ti + (lane & 15)) * KT + 32 * g + 8 * (lane >> 4) + e]; b[e] = B[(16 * tj + (lane & 15)) * KT + 32 * g + 8 * (lane >> 4) + e]; }
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
            }
            for (int r = 0; r < 4; ++r) c16[(16 * ti + 4 * (lane >> 4) + r) * 32 + 16 * tj + (lane & 15)] = acc[r];
        }
}

int main() {
    std::vector<_Float16> A(32 * KT), B(32 * KT);
    srand(1);
    int differ_total = 0;
    for (int trial = 0; trial < 50; ++trial) {
        for (int i = 0; i < 32 * KT; ++i) {
            const float m = (rand() / (float)RAND_MAX - 0.5f), e = (float)(1 << (rand() % 12));
            A[i] = (_Float16)(m * e); B[i] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * (1 << (rand() % 12)) / 64.f);
        }
        _Float16 *dA, *dB; float *d32, *d16;
        hipMalloc(&dA, 64 * KT); hipMalloc(&dB, 64 * KT); hipMalloc(&d32, 4096); hipMalloc(&d16, 4096);
        hipMemcpy(dA, A.data(), 64 * KT, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 64 * KT, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, d32, d16);
        std::vector<float> c32(1024), c16(1024);
        hipMemcpy(c32.data(), d32, 4096, hipMemcpyDeviceToHost); hipMemcpy(c16.data(), d16, 4096, hipMemcpyDeviceToHost);
        int differ = 0; double maxrel = 0;
        for (int i = 0; i < 1024; ++i) if (memcmp(&c32[i], &c16[i], 4)) { ++differ; double r = fabs((double)c32[i] - c16[i]) / (fabs((double)c32[i]) + 1e-30); if (r > maxrel) maxrel = r; }
        differ_total += differ;
        if (trial < 3 || differ) printf("trial %d: %d of 1024 outputs differ (max rel %.3g)\n", trial, differ, maxrel);
        hipFree(dA); hipFree(dB); hipFree(d32); hipFree(d16);
    }
    printf("total differing outputs over 50 trials: %d\n", differ_total);
    return 0;
}
