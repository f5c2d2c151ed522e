#include <hip/hip_runtime.h>
__global__ void k(const float *src, float *dst, unsigned n) {
    __shared__ __attribute__((aligned(16))) float buf[64 * 4 * 2];
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, n * 4u, 0x00020000);
    unsigned off = (threadIdx.x * 7u % 64u) * 16u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)buf, 16, off, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)(buf + 256), 16, threadIdx.x == 3 ? 0x80000000u : off + 4u, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) dst[i] = buf[i];
}
int main() {
    float *s, *d; hipMalloc(&s, 4096 * 4); hipMalloc(&d, 512 * 4);
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = i;
    hipMemcpy(s, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, 1, 64, 0, 0, s, d, 4096u);
    float o[512]; hipMemcpy(o, d, sizeof o, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
        float want0 = (l * 7 % 64) * 4 + e, want1 = l == 3 ? 0 : (l * 7 % 64) * 4 + 1 + e;
        if (o[l * 4 + e] != want0) { if (bad < 5) printf("A lane %d e %d got %g want %g\n", l, e, o[l*4+e], want0); ++bad; }
        if (o[256 + l * 4 + e] != want1) { if (bad < 5) printf("B lane %d e %d got %g want %g\n", l, e, o[256+l*4+e], want1); ++bad; }
    }
    printf("dma16: %d mismatches (B = dword-aligned, not 16-byte aligned source; lane 3 out of range -> zeros)\n", bad);
    return bad != 0;
}
