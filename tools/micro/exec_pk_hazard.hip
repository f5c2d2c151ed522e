// Hypothesis for the chain race (profiles/r5q_chain_race.txt): a packed-fp32 VALU instruction issued right behind a SALU write of EXEC
// (the `s_or_b64 exec, exec, sN` that closes a masked region) executes its LAST lane quarter (lanes 48..63) under the OLD mask when the
// wave's issue timing is perturbed by other processes' waves -- those lanes keep their old register contents.  Each thread: a masked
// region (odd lanes only) with a dependent global load, EXEC restored, IMMEDIATELY a v_pk_add_f32 on registers whose expected result
// is known; mismatches are counted per lane quarter.  Variants: no wait state, with s_nop 0..4 between the EXEC write and the packed op,
// and a plain v_add_f32 pair instead of the packed op.
//   hipcc --offload-arch=gfx950 -O3 exec_pk_hazard.hip -o exec_pk_hazard && ./exec_pk_hazard [launches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int V>
__global__ __launch_bounds__(256) void k(const float *__restrict__ in, int n, int rounds, unsigned long long *bad, unsigned *quarters) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    f2 acc = {0.f, 0.f};
    float s0 = 0.f, s1 = 0.f;
    for (int r = 0; r < rounds; ++r) {
        const f2 add = {(float)(r + 1), (float)(2 * r + 1)};
        float ld = 0.f, tmp;
        unsigned long long saved;
        const float *ptr = in + ((id * 7 + r * 13) % n);
        // ONE assembly block (the compiler must never run its own code under the narrowed mask): odd lanes load through a dependent
        // address, EXEC is restored, and the packed (or plain) add follows with V's wait states in between
#define HEAD "v_and_b32 %[t], 1, %[tid]\n\tv_cmp_eq_u32 vcc, 1, %[t]\n\ts_and_saveexec_b64 %[sv], vcc\n\tglobal_load_dword %[ld], %[addr], off\n\ts_waitcnt vmcnt(0)\n\ts_or_b64 exec, exec, %[sv]\n\t"
        if (V == 0)      asm volatile(HEAD "v_pk_add_f32 %[acc], %[acc], %[add]" : [acc] "+v"(acc), [t] "=&v"(tmp), [sv] "=&s"(saved), [ld] "=&v"(ld) : [tid] "v"(threadIdx.x), [addr] "v"(ptr), [add] "v"(add) : "vcc", "memory");
        else if (V == 1) asm volatile(HEAD "s_nop 0\n\tv_pk_add_f32 %[acc], %[acc], %[add]" : [acc] "+v"(acc), [t] "=&v"(tmp), [sv] "=&s"(saved), [ld] "=&v"(ld) : [tid] "v"(threadIdx.x), [addr] "v"(ptr), [add] "v"(add) : "vcc", "memory");
        else if (V == 2) asm volatile(HEAD "s_nop 4\n\tv_pk_add_f32 %[acc], %[acc], %[add]" : [acc] "+v"(acc), [t] "=&v"(tmp), [sv] "=&s"(saved), [ld] "=&v"(ld) : [tid] "v"(threadIdx.x), [addr] "v"(ptr), [add] "v"(add) : "vcc", "memory");
        else             asm volatile(HEAD "v_add_f32 %[a0], %[a0], %[x]\n\tv_add_f32 %[a1], %[a1], %[y]" : [a0] "+v"(s0), [a1] "+v"(s1), [t] "=&v"(tmp), [sv] "=&s"(saved), [ld] "=&v"(ld) : [tid] "v"(threadIdx.x), [addr] "v"(ptr), [x] "v"(add.x), [y] "v"(add.y) : "vcc", "memory");
        if (ld == -12345.f) acc.x += 1.f;
    }
    float e0 = 0.f, e1 = 0.f;
    for (int r = 0; r < rounds; ++r) { e0 += (float)(r + 1); e1 += (float)(2 * r + 1); }
    const bool diff = V == 3 ? (s0 != e0 || s1 != e1) : (acc.x != e0 || acc.y != e1);
    if (diff) { atomicAdd(bad, 1ull); atomicOr(quarters + ((threadIdx.x & 63) >> 4), 1u); }
}

template <int V>
static void run(const char *name, int launches, const float *in, int n) {
    unsigned long long *bad, hb = 0; unsigned *q, hq[4] = {0, 0, 0, 0};
    hipMalloc(&bad, 8); hipMalloc(&q, 16); hipMemset(bad, 0, 8); hipMemset(q, 0, 16);
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(k<V>, dim3(2048), dim3(256), 0, 0, in, n, 32, bad, q);
    hipDeviceSynchronize();
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(hq, q, 16, hipMemcpyDeviceToHost);
    printf("%-52s %d launches: %llu lanes wrong (lane quarters hit: %u %u %u %u)\n", name, launches, hb, hq[0], hq[1], hq[2], hq[3]);
}

int main(int argc, char **argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 2000, n = 1 << 22;
    float *d; hipMalloc(&d, (size_t)n * 4); hipMemset(d, 0, (size_t)n * 4);
    run<0>("s_or exec; v_pk_add_f32 (no wait state)", launches, d, n);
    run<1>("s_or exec; s_nop 0; v_pk_add_f32", launches, d, n);
    run<2>("s_or exec; s_nop 4; v_pk_add_f32", launches, d, n);
    run<3>("s_or exec; v_add_f32 x 2 (no packed op)", launches, d, n);
    return 0;
}
