// Do kernel arguments beyond the first 256 bytes reach every wave of a launch?  A kernel takes a struct of N 8-byte words by
// value (N * 8 bytes of kernarg) holding the launch number in every word, and every thread checks every word against the launch
// number passed in the FIRST word; mismatches are counted.  Run it beside other GPU processes (tools/race_kernels.py --load-seconds N):
// with the GPU to itself nothing ever mismatches.
//   hipcc --offload-arch=gfx950 -O3 kernarg_race.hip -o kernarg_race && ./kernarg_race [launches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int N> struct Args { unsigned long long w[N]; };

template <int N>
__global__ void k(Args<N> a, unsigned long long *bad, unsigned long long *where) {
    const unsigned long long want = a.w[0];
    unsigned long long first = ~0ull;
#pragma unroll
    for (int i = 1; i < N; ++i)
        if (a.w[i] != want && first == ~0ull) first = ((unsigned long long)i << 48) | (a.w[i] & 0xffffffffffffull);
    if (first != ~0ull) { atomicAdd(bad, 1ull); *where = first; }
}

template <int N>
static void run(int launches) {
    unsigned long long *bad, *where, h[2] = {0, 0};
    hipMalloc(&bad, 8); hipMalloc(&where, 8);
    hipMemset(bad, 0, 8); hipMemset(where, 0, 8);
    unsigned long long bad_launches = 0, prev = 0;
    for (int l = 1; l <= launches; ++l) {
        Args<N> a;
        for (int i = 0; i < N; ++i) a.w[i] = (unsigned long long)l;
        hipLaunchKernelGGL(k<N>, dim3(1024), dim3(256), 0, 0, a, bad, where);
        if (l % 64 == 0) {
            hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
            if (h[0] != prev) { ++bad_launches; prev = h[0]; }
        }
    }
    hipDeviceSynchronize();
    hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(h + 1, where, 8, hipMemcpyDeviceToHost);
    printf("kernarg %4d bytes: %d launches, %llu threads saw a stale word (last: word %llu held launch %llu)\n", N * 8 + 16, launches, h[0],
           h[1] >> 48, h[1] & 0xffffffffffffull);
    hipFree(bad); hipFree(where);
}

int main(int argc, char **argv) {
    const int launches = argc > 1 ? atoi(argv[1]) : 20000;
    run<8>(launches); run<24>(launches); run<30>(launches); run<34>(launches); run<40>(launches); run<60>(launches); run<120>(launches);
    return 0;
}
