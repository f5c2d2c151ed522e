// How fast can a CU pull L2-resident data into LDS?  (a) LDS-DMA (buffer_load_dwordx4 ... lds), (b) buffer_load_dwordx4
// into registers + ds_write_b128.  One workgroup of W waves per CU, each wave streams its own 8-row x 128-byte pieces
// of a small (L2-resident) buffer round and round; no compute.   hipcc --offload-arch=gfx950 -O3 lds_fill.hip -o lds_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// GEMM-like staging: one instruction = 8 rows x 128 bytes, rows `stride` bytes apart (a 32-channel slab of 8 cells of a
// pixel-major activation map); successive instructions of a wave move on by 8 rows
template <int DEPTH>
__global__ __launch_bounds__(512) void fill_rows_kernel(const float *src, unsigned bytes, unsigned stride, int iters, float *sink) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, bytes, 0x00020000);
    const unsigned waves = blockDim.x >> 6;
    const unsigned rows = bytes / stride;
    unsigned row = (blockIdx.x * 131u + wid * 8u + (lane >> 3)) % rows;
    unsigned col = 0;
    float *dst = lds + wid * DEPTH * 256;
    f32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int p = 0; p < DEPTH; ++p) {
            const unsigned off = row * stride + col + (lane & 7) * 16u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(dst + p * 256), 16, off, 0, 0, 0);
            row += waves * 8u; if (row >= rows) { row -= rows; col += 128u; if (col + 128u > stride) col = 0; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((i & 63) == 63) acc += *reinterpret_cast<f32x4 *>(dst + lane * 4);
    }
    if (acc.x == 12345.f) sink[0] = acc.y;
}

template <int DEPTH>
static void run_rows(int waves, unsigned stride, const float *src, unsigned bytes, float *sink) {
    const int iters = 2000, cus = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = (size_t)waves * DEPTH * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void *>(fill_rows_kernel<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((fill_rows_kernel<DEPTH>), dim3(cus), dim3(64 * waves), lds, 0, src, bytes, stride, iters, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double total = (double)cus * waves * DEPTH * 1024.0 * iters;
    printf("LDS-DMA rows, stride %5u B, buffer %3u MiB, waves %d depth %d: %7.1f GB/s per CU, %6.2f TB/s chip\n", stride, bytes >> 20, waves, DEPTH,
           total / cus / (ms * 1e-3) / 1e9, total / (ms * 1e-3) / 1e12);
}

template <int MODE, int DEPTH>
__global__ __launch_bounds__(512) void fill_kernel(const float *src, unsigned bytes, int iters, float *sink) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, bytes, 0x00020000);
    // each wave: DEPTH pieces of 1 KiB in flight; piece p of iteration i at byte offset ((i * DEPTH + p) * waves + wid) * 1024 (mod bytes)
    const unsigned waves = blockDim.x >> 6;
    unsigned off = (blockIdx.x * 7919u * 1024u + wid * 1024u + lane * 16u) % bytes;
    float *dst = lds + wid * DEPTH * 256;
    f32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int p = 0; p < DEPTH; ++p) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(dst + p * 256), 16, off, 0, 0, 0);
                off += waves * 1024u; if (off >= bytes) off -= bytes;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            f32x4 v[DEPTH];
#pragma unroll
            for (int p = 0; p < DEPTH; ++p) {
                v[p] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
                off += waves * 1024u; if (off >= bytes) off -= bytes;
            }
#pragma unroll
            for (int p = 0; p < DEPTH; ++p) *reinterpret_cast<f32x4 *>(dst + p * 256 + lane * 4) = v[p];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if ((i & 63) == 63) acc += *reinterpret_cast<f32x4 *>(dst + lane * 4);     // keep the LDS contents alive
    }
    if (acc.x == 12345.f) sink[0] = acc.y;
}

template <int MODE, int DEPTH>
static void run(const char *name, int waves, const float *src, unsigned bytes, float *sink) {
    const int iters = 2000, cus = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = (size_t)waves * DEPTH * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void *>(fill_kernel<MODE, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((fill_kernel<MODE, DEPTH>), dim3(cus), dim3(64 * waves), lds, 0, src, bytes, iters, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double total = (double)cus * waves * DEPTH * 1024.0 * iters;
    printf("%-28s waves %d depth %d: %7.1f GB/s per CU, %6.2f TB/s chip\n", name, waves, DEPTH, total / cus / (ms * 1e-3) / 1e9, total / (ms * 1e-3) / 1e12);
}

int main() {
    const unsigned bytes = 2u << 20;                    // 2 MiB: stays in every XCD's L2
    float *src, *sink;
    hipMalloc(&src, bytes); hipMalloc(&sink, 64);
    hipMemset(src, 0, bytes);
    for (int waves : {4, 8}) {
        if (waves == 4) { run<0, 4>("LDS-DMA dwordx4", 4, src, bytes, sink); run<0, 8>("LDS-DMA dwordx4", 4, src, bytes, sink); run<1, 4>("VGPR dwordx4 + ds_write_b128", 4, src, bytes, sink); run<1, 8>("VGPR dwordx4 + ds_write_b128", 4, src, bytes, sink); }
        else { run<0, 4>("LDS-DMA dwordx4", 8, src, bytes, sink); run<0, 8>("LDS-DMA dwordx4", 8, src, bytes, sink); run<1, 4>("VGPR dwordx4 + ds_write_b128", 8, src, bytes, sink); run<1, 8>("VGPR dwordx4 + ds_write_b128", 8, src, bytes, sink); }
    }
    // strided rows, L2-resident (2 MiB) and not (64 MiB: MALL / HBM)
    const unsigned big = 64u << 20;
    float *src2; hipMalloc(&src2, big); hipMemset(src2, 0, big);
    for (unsigned stride : {128u, 512u, 1024u, 1536u, 2848u, 5120u}) run_rows<4>(8, stride, src, bytes, sink);
    for (unsigned stride : {128u, 1024u, 1536u}) run_rows<4>(8, stride, src2, big, sink);
    for (unsigned stride : {1024u, 1536u}) run_rows<8>(4, stride, src2, big, sink);
    return 0;
}
