// What does the chip's power limit leave of the fp16 matrix peak?  A bare loop of v_mfma_f32_32x32x16_f16 -- no LDS, no memory,
// operands in registers -- on every CU, with operands that are all zero, constant, or random (fp16 normals; or the split
// arithmetic's pairs: hi = fp16(x), lo = fp16((x - hi) * 2048)), rotating through 8 register sets so that the pipe's inputs
// toggle like a GEMM's.  Per variant: the shader clock MEASURED over the last quarter of the run (s_memtime ticks per
// s_memrealtime tick of 10 ns), MFMAs per SIMD and microsecond, TF/s.  The run is ~20 ms long: the power controller has settled.
//   hipcc --offload-arch=gfx950 -O3 mfma_power.hip -o mfma_power && ./mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void k(const f16x8 *__restrict__ ops, float *out, int iters, unsigned long long *stamps) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    f16x8 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = ops[(i * 2) * 64 + lane]; b[i] = ops[(i * 2 + 1) * 64 + lane]; }
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    unsigned long long t[3], r[3];
    auto stamp = [&](int s) { asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t[s]), "=s"(r[s]) :: "memory"); };
    stamp(0);
    for (int it = 0; it < iters; ++it) {
        if (it == iters - iters / 4) stamp(1);
#pragma unroll
        for (int rot = 0; rot < 8; ++rot)
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(m + rot) & 7], b[(m + 3 * rot) & 7], acc[m], 0, 0, 0);
    }
    stamp(2);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) for (int r2 = 0; r2 < 16; ++r2) s += acc[i][r2];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0 && wv == 0) { for (int i = 0; i < 3; ++i) { stamps[blockIdx.x * 6 + i] = t[i]; stamps[blockIdx.x * 6 + 3 + i] = r[i]; } }
}

static unsigned short f16bits(float x) { _Float16 h = (_Float16)x; unsigned short u; memcpy(&u, &h, 2); return u; }

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    f16x8 *d_ops; float *d_out; unsigned long long *d_st;
    hipMalloc(&d_ops, 16 * 64 * 16); hipMalloc(&d_out, 1024 * 512 * 4); hipMalloc(&d_st, 1024 * 6 * 8);
    std::vector<unsigned long long> st(1024 * 6);
    const char *names[] = {"all zero", "constant 1.0", "random fp16 normals", "split pairs (hi, lo) of random fp32"};
    printf("# %s, %d CUs; v_mfma_f32_32x32x16_f16 alone, 8 accumulators, operands rotating through 8 register sets\n", prop.name, cus);
    for (int kind = 0; kind < 4; ++kind) {
        std::vector<unsigned short> h(16 * 64 * 8);
        srand(1);
        auto nrm = [] { float u = (rand() + 1.f) / (RAND_MAX + 2.f), v = (rand() + 1.f) / (RAND_MAX + 2.f); return sqrtf(-2.f * logf(u)) * cosf(6.2831853f * v); };
        for (size_t i = 0; i < h.size(); ++i) {
            float x = nrm();
            if (kind == 0) h[i] = 0;
            else if (kind == 1) h[i] = f16bits(1.f);
            else if (kind == 2) h[i] = f16bits(x);
            else { const size_t set = i / (64 * 8); _Float16 hi = (_Float16)x; h[i] = (set & 2) ? f16bits((x - (float)hi) * 2048.f) : f16bits(x); }   // half of the sets hold low halves
        }
        hipMemcpy(d_ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        for (int waves : {4, 8})
            for (int wgs : {cus * 7 / 8, cus}) {
                const int iters = waves == 8 ? 6000 : 12000;     // ~20 ms
                hipLaunchKernelGGL(k, dim3(wgs), dim3(64 * waves), 0, 0, d_ops, d_out, iters, d_st);
                hipDeviceSynchronize();
                hipMemcpy(st.data(), d_st, wgs * 6 * 8, hipMemcpyDeviceToHost);
                double clk = 0, span = 0;
                for (int w = 0; w < wgs; ++w) {
                    clk += (double)(st[w * 6 + 2] - st[w * 6 + 1]) / (double)(st[w * 6 + 5] - st[w * 6 + 4]) * 0.1;
                    span += (double)(st[w * 6 + 5] - st[w * 6 + 4]) * 0.01;      // us, last quarter
                }
                clk /= wgs; span /= wgs;
                const double mfmas = (double)(iters / 4) * 64.0 * waves;          // per workgroup in the last quarter
                const double tf = mfmas * wgs * 32768.0 / (span * 1e-6) / 1e12;
                printf("%-38s %d waves/SIMD, %3d workgroups: clock %.2f GHz, %.1f cycles per MFMA and SIMD, %6.0f TF/s (%.2f of 2500)\n", names[kind], waves / 4, wgs, clk,
                       clk * 1e3 * span / (mfmas / 4.0), tf, tf / 2500.0);
            }
    }
    return 0;
}
