// Where does the power go in a tile-resident GEMM step?  The chip is power-limited under dense matrix work on random data
// (tools/micro/mfma_power.hip: a bare MFMA loop runs at ~1.6-1.7 GHz instead of 2.4).  This loop is one K step of
// tile_conv_kernel -- 12 x v_mfma_f32_32x32x16_f16 on 4 + 4 A fragments (hi, lo) and one B pair -- with the step's operand
// traffic switched on piece by piece: A fragments re-read from LDS every step (8 ds_read_b128) or held in registers, B
// fragments streamed from an L2-resident buffer every step (2 global_load_dwordx4) or held.  One wave per SIMD (the arbiter
// serves two waves unevenly; one wave shows the rates cleanly), every CU busy, random fp16 operands.  Reported: the MEASURED
// shader clock over the last quarter of a ~20 ms run, cycles per MFMA, TF/s.  Clock x utilisation is what the power budget buys.
//   hipcc --offload-arch=gfx950 -O3 mfma_energy.hip -o mfma_energy && ./mfma_energy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int LDSR, int GLD, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(const uint4 *__restrict__ wbuf, const uint4 *__restrict__ abuf, float *out, int iters, unsigned long long *stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // 128 KB of random "activations" in LDS
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) reinterpret_cast<uint4 *>(lds)[i] = abuf[i];
    __syncthreads();
    f32x16 acc[4], accx[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; accx[i][r] = 0.f; }
    uint4 ah[4], al[4], bh, bl;
    const unsigned char *ab = lds + lane * 16;
    for (int i = 0; i < 4; ++i) { ah[i] = *reinterpret_cast<const uint4 *>(ab + i * 2048); al[i] = *reinterpret_cast<const uint4 *>(ab + i * 2048 + 1024); }
    const uint4 *wp = wbuf + lane;      // one 1.3 MB stream shared by all workgroups (L2-resident), 164 KB of it per wave
    bh = wp[0]; bl = wp[64];
    unsigned long long t[3], r[3];
    auto stamp = [&](int s) { asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t[s]), "=s"(r[s]) :: "memory"); };
    stamp(0);
    for (int it = 0; it < iters; ++it) {
        if (it == iters - iters / 4) stamp(1);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int step = (it * 8 + s) % 80;                  // 80 steps of 2 KB = 164 KB per wave, 1.3 MB per workgroup of 8: the z | r gates' weights
            if (LDSR) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ah[i] = *reinterpret_cast<const uint4 *>(ab + ((step & 7) * 8 + i * 2) * 1024 + wv * 0);
                    al[i] = *reinterpret_cast<const uint4 *>(ab + ((step & 7) * 8 + i * 2 + 1) * 1024);
                }
            }
            if (GLD) { bh = wp[(size_t)step * 128 + (size_t)wv * 10240]; bl = wp[(size_t)step * 128 + 64 + (size_t)wv * 10240]; }
            const f16x8 wh = __builtin_bit_cast(f16x8, bh), wl = __builtin_bit_cast(f16x8, bl);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, __builtin_bit_cast(f16x8, ah[i]), acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) accx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, __builtin_bit_cast(f16x8, ah[i]), accx[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) accx[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, __builtin_bit_cast(f16x8, al[i]), accx[i], 0, 0, 0);
            if (!LDSR) {      // held operands still rotate, so that the pipe's inputs toggle
                uint4 t0 = ah[0]; ah[0] = ah[1]; ah[1] = ah[2]; ah[2] = ah[3]; ah[3] = t0;
                t0 = al[0]; al[0] = al[1]; al[1] = al[2]; al[2] = al[3]; al[3] = t0;
            }
        }
    }
    stamp(2);
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r2 = 0; r2 < 16; ++r2) s += acc[i][r2] + accx[i][r2];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0 && wv == WAVES - 1) { for (int i = 0; i < 3; ++i) { stamps[blockIdx.x * 6 + i] = t[i]; stamps[blockIdx.x * 6 + 3 + i] = r[i]; } }
}

static unsigned short f16bits(float x) { _Float16 h = (_Float16)x; unsigned short u; memcpy(&u, &h, 2); return u; }

template <int LDSR, int GLD, int WAVES>
static void run(const char *name, int wgs, const uint4 *d_w, const uint4 *d_a, float *d_out, unsigned long long *d_st) {
    std::vector<unsigned long long> st(1024 * 6);
    const int iters = WAVES == 8 ? 8000 : 16000;
    auto kern = k<LDSR, GLD, WAVES>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(64 * WAVES), 131072, 0, d_w, d_a, d_out, iters, d_st);
    hipDeviceSynchronize();
    hipMemcpy(st.data(), d_st, wgs * 6 * 8, hipMemcpyDeviceToHost);
    double clk = 0, span = 0;
    for (int w = 0; w < wgs; ++w) {
        clk += (double)(st[w * 6 + 2] - st[w * 6 + 1]) / (double)(st[w * 6 + 5] - st[w * 6 + 4]) * 0.1;
        span += (double)(st[w * 6 + 5] - st[w * 6 + 4]) * 0.01;
    }
    clk /= wgs; span /= wgs;
    const double mfmas = (double)(iters / 4) * 96.0 * WAVES;
    const double tf = mfmas * wgs * 32768.0 / (span * 1e-6) / 1e12;
    printf("%-58s %d waves/SIMD, %3d workgroups: clock %.2f GHz, %5.1f cycles per MFMA and SIMD, %6.0f TF/s (%.2f of 2500)\n", name, WAVES / 4, wgs, clk,
           clk * 1e3 * span / (mfmas / 4.0), tf, tf / 2500.0);
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    uint4 *d_w, *d_a; float *d_out; unsigned long long *d_st;
    const size_t wbytes = (size_t)16 << 20;
    hipMalloc(&d_w, wbytes); hipMalloc(&d_a, 131072); hipMalloc(&d_out, 1024 * 512 * 4); hipMalloc(&d_st, 1024 * 6 * 8);
    std::vector<unsigned short> h(wbytes / 2);
    srand(1);
    auto nrm = [] { float u = (rand() + 1.f) / (RAND_MAX + 2.f), v = (rand() + 1.f) / (RAND_MAX + 2.f); return sqrtf(-2.f * logf(u)) * cosf(6.2831853f * v); };
    for (size_t i = 0; i < h.size(); ++i) h[i] = f16bits(nrm() * 0.05f);
    hipMemcpy(d_w, h.data(), wbytes, hipMemcpyHostToDevice);
    for (size_t i = 0; i < 65536; ++i) h[i] = f16bits(nrm());
    hipMemcpy(d_a, h.data(), 131072, hipMemcpyHostToDevice);
    printf("# %d CUs; one K step of the tile-resident GEMM = 12 MFMAs; operand traffic switched on piece by piece; random fp16 operands\n", cus);
    for (int wgs : {cus * 7 / 8, cus}) {
        run<0, 0, 4>("operands held in registers", wgs, d_w, d_a, d_out, d_st);
        run<1, 0, 4>("+ 8 ds_read_b128 per step (A fragments from LDS)", wgs, d_w, d_a, d_out, d_st);
        run<0, 1, 4>("+ 2 global_load_dwordx4 per step (B fragments from L2)", wgs, d_w, d_a, d_out, d_st);
        run<1, 1, 4>("+ both (the kernel's K loop)", wgs, d_w, d_a, d_out, d_st);
        run<1, 1, 8>("+ both, two waves per SIMD (last wave timed)", wgs, d_w, d_a, d_out, d_st);
    }
    return 0;
}
