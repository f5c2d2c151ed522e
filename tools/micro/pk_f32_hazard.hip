// Which packed-fp32 instruction of gfx950 loses lanes 48-63 when other processes share the GPU?  (Found through MFT's chain + select
// kernels: 16 wrong pixels -- lanes 48..63 of one wave, ONE output register -- in a third of the launches under contention, never with
// the GPU to itself; gone with -fno-slp-vectorize, i.e. without v_pk_*_f32 / v_pk_mov_b32.)  Each variant runs a dependent chain of ONE
// kind of packed instruction (with the one wait state the compiler puts between them) next to the same arithmetic in scalar
// instructions, and counts lanes whose results differ.  Run beside other GPU processes.
//   hipcc --offload-arch=gfx950 -O3 pk_f32_hazard.hip -o pk_f32_hazard && ./pk_f32_hazard [launches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int V>
__global__ __launch_bounds__(256) void k(const float *__restrict__ in, int n, int rounds, unsigned long long *bad, unsigned *lanes) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    f2 a = {in[id % n], in[(id * 7 + 1) % n]}, b = {in[(id * 3 + 2) % n], in[(id * 5 + 3) % n]};
    f2 p = a, q = b;                 // packed path
    float s0 = a.x, s1 = a.y, t0 = b.x, t1 = b.y;   // scalar path
    for (int r = 0; r < rounds; ++r) {
        if (V == 0) {          // v_pk_add_f32 / v_pk_mul_f32, plain
            asm volatile("v_pk_add_f32 %0, %0, %1\n\ts_nop 0\n\tv_pk_mul_f32 %0, %0, 0.5 op_sel_hi:[1,0]\n\ts_nop 0" : "+v"(p) : "v"(q));
            asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\tv_mul_f32 %0, 0.5, %0\n\tv_mul_f32 %1, 0.5, %1" : "+v"(s0), "+v"(s1) : "v"(t0), "v"(t1));
        } else if (V == 1) {   // op_sel / neg modifiers: p = (p.y - q.x, p.x - q.y) ... then halved
            asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 0\n\tv_pk_mul_f32 %0, %0, 0.5 op_sel_hi:[1,0]\n\ts_nop 0" : "+v"(p) : "v"(q));
            float n0, n1;
            asm volatile("v_sub_f32 %0, %3, %4\n\tv_sub_f32 %1, %2, %5\n\tv_mul_f32 %0, 0.5, %0\n\tv_mul_f32 %1, 0.5, %1" : "=&v"(n0), "=&v"(n1) : "v"(s0), "v"(s1), "v"(t0), "v"(t1));
            s0 = n0; s1 = n1;
        } else {               // v_pk_mov_b32: p = (p.y, q.x); q = (q.y, p_old.x) ... a register shuffle
            f2 np_, nq_;
            asm volatile("v_pk_mov_b32 %0, %2, %3 op_sel:[1,0]\n\tv_pk_mov_b32 %1, %3, %2 op_sel:[1,0]\n\ts_nop 0" : "=&v"(np_), "=&v"(nq_) : "v"(p), "v"(q));
            p = np_; q = nq_;
            const float o0 = s0, o1 = s1;
            s0 = o1; s1 = t0; t0 = t1; t1 = o0;
            asm volatile("" : "+v"(s0), "+v"(s1), "+v"(t0), "+v"(t1));
        }
    }
    const bool diff = V == 2 ? (p.x != s0 || p.y != s1 || q.x != t0 || q.y != t1) : (p.x != s0 || p.y != s1);
    if (diff) { atomicAdd(bad, 1ull); atomicOr(lanes + ((threadIdx.x & 63) >> 4), 1u); }
}

template <int V>
static void run(const char *name, int launches, const float *in, int n) {
    unsigned long long *bad, hb = 0; unsigned *lanes, hl[4] = {0, 0, 0, 0};
    hipMalloc(&bad, 8); hipMalloc(&lanes, 16); hipMemset(bad, 0, 8); hipMemset(lanes, 0, 16);
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(k<V>, dim3(2048), dim3(256), 0, 0, in, n, 64, bad, lanes);
    hipDeviceSynchronize();
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(hl, lanes, 16, hipMemcpyDeviceToHost);
    printf("%-44s %d launches: %llu lanes differ from the scalar path (lane quarters hit: %u %u %u %u)\n", name, launches, hb, hl[0], hl[1], hl[2], hl[3]);
}

int main(int argc, char **argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 3000, n = 1 << 20;
    float *h = (float *)malloc(n * 4), *d;
    unsigned s = 1;
    for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) / 16777216.f * 4.f - 2.f; }
    hipMalloc(&d, n * 4); hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    run<0>("v_pk_add_f32 / v_pk_mul_f32", launches, d, n);
    run<1>("v_pk_add_f32 with op_sel + neg modifiers", launches, d, n);
    run<2>("v_pk_mov_b32", launches, d, n);
    return 0;
}
