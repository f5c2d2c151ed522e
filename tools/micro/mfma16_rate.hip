// Can ONE wave per SIMD keep the matrix pipe busy with v_mfma_f32_16x16x32_f16 (4 passes, half a 32x32x16)?  ns per
// MFMA for a loop of 24 independent accumulators, bare and with the 16-row conv tile's companions: one ds_read_b128 per
// 2.5 MFMAs, one LDS-DMA piece per 7.
//   hipcc --offload-arch=gfx950 -O3 mfma16_rate.hip -o mfma16_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int SHAPE>
__global__ __launch_bounds__(256) void k(float *out, const float *in, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    const int lane = threadIdx.x & 63;
    f32x4 acc[24];
    f32x16 big[6];
    for (int i = 0; i < 24; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) big[i][r] = 0.f;
    f16x8 a[3], b[4];
    for (int o = 0; o < 3; ++o) for (int e = 0; e < 8; ++e) a[o][e] = (_Float16)(lane + e + o);
    for (int o = 0; o < 4; ++o) for (int e = 0; e < 8; ++e) b[o][e] = (_Float16)(lane - e - o);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(in), 0, 1u << 20, 0x00020000);
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    f16x8 ld = a[0];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            if (SHAPE == 16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a[m % 3]), "v"(b[m % 4]));
            else if (m % 4 == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(big[m / 4]) : "v"(a[m % 3]), "v"(b[m % 4]));   // 6 per 24 slots: same flops per slot pair... (half)
            __builtin_amdgcn_sched_barrier(0);
            if ((KIND & 1) && m % 5 < 2) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"((unsigned)(lane * 16 + m * 1024)));
                __builtin_amdgcn_sched_barrier(0);
            }
            if ((KIND & 2) && m % 7 == 3) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(lds + 8192 + (threadIdx.x >> 6) * 256), 16,
                                                         (unsigned)(lane * 16 + (it & 63) * 1024 + m * 16384), 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (KIND) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    float s = (float)ld[0];
    for (int i = 0; i < 24; ++i) s += acc[i][0];
    for (int i = 0; i < 6; ++i) s += big[i][0];
    if (s == 123.456f) out[0] = s;
}

template <int KIND, int SHAPE>
static void run(const char *name, float *out, float *in) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND, SHAPE>), dim3(256), dim3(256), 0, 0, out, in, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double n = SHAPE == 16 ? 24.0 : 6.0, fl = SHAPE == 16 ? 16384.0 : 32768.0;
    printf("%-44s %6.2f ns per MFMA = %5.0f TF chip (1 wave per SIMD)\n", name, ms * 1e6 / (iters * n), fl * iters * n * 1024 / (ms * 1e-3) / 1e12);
}

int main() {
    float *out, *in;
    hipMalloc(&out, 1024); hipMalloc(&in, 8u << 20); hipMemset(in, 0, 8u << 20);
    run<0, 16>("16x16x32 bare", out, in);
    run<0, 32>("32x32x16 bare", out, in);
    run<1, 16>("16x16x32 + ds_read_b128 per 2.5", out, in);
    run<2, 16>("16x16x32 + LDS-DMA piece per 7", out, in);
    run<3, 16>("16x16x32 + both", out, in);
    return 0;
}
