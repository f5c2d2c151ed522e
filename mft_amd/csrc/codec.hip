// Flow-cache codec on the device: the ".flowouX16" quantisation of
// MFT/utils/io.py:495-512 (write) and :548-551 (read), per channel:
//   lb = min(x), ub = max(x)
//   q  = uint16(round_half_even( (x - lb) / (ub - lb) * 65535 ))     (all zeros if |ub - lb| < 1e-8)
//   x' = (float(q) / 65535) * (ub - lb) + lb
// in float32, operation by operation as numpy does it (compiled with -ffp-contract=off).
// Quantising on the device halves the bytes that cross PCIe when a cache entry is
// spilled to disk; the PNG container around the uint16 planes is host work
// (mft_amd/flowou_codec.py), with the one inherently serial step -- PNG row
// unfiltering -- in C below.
//
// HBM-bound: 4 B/element read twice + 2 B written (min/max pass, quantise pass);
// the second pass re-reduces the per-block partials instead of a third launch.
#include "common.h"
#include "profile.h"
#include <cfloat>
#include <cstdlib>

namespace mftx {

constexpr int QT = 256;          // threads per block
constexpr int QB_MAX = 1024;     // blocks (= partial min/max pairs) at most

__device__ __forceinline__ void wave_minmax(float &lo, float &hi) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off));
        hi = fmaxf(hi, __shfl_xor(hi, off));
    }
}

__device__ __forceinline__ void block_minmax(float &lo, float &hi, float *sh /* [2 * QT / 64] */) {
    wave_minmax(lo, hi);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[2 * w] = lo; sh[2 * w + 1] = hi; }
    __syncthreads();
    lo = sh[0]; hi = sh[1];
#pragma unroll
    for (int i = 1; i < QT / 64; ++i) { lo = fminf(lo, sh[2 * i]); hi = fmaxf(hi, sh[2 * i + 1]); }
    __syncthreads();
}

__global__ __launch_bounds__(QT) void minmax_partial_kernel(const float *__restrict__ x, long long n,
                                                            float *__restrict__ partial) {
    __shared__ float sh[2 * QT / 64];
    float lo = FLT_MAX, hi = -FLT_MAX;
    const long long n4 = n >> 2;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    for (long long i = (long long)blockIdx.x * QT + threadIdx.x; i < n4; i += (long long)gridDim.x * QT) {
        const float4 v = x4[i];
        lo = fminf(fminf(lo, v.x), fminf(v.y, fminf(v.z, v.w)));
        hi = fmaxf(fmaxf(hi, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
    }
    if (blockIdx.x == 0)
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += QT) { lo = fminf(lo, x[i]); hi = fmaxf(hi, x[i]); }
    block_minmax(lo, hi, sh);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = lo; partial[2 * blockIdx.x + 1] = hi; }
}

__global__ __launch_bounds__(QT) void quantize_kernel(const float *__restrict__ x, long long n,
                                                      const float *__restrict__ partial, int n_partial,
                                                      unsigned short *__restrict__ q, float *__restrict__ lohi) {
    __shared__ float sh[2 * QT / 64];
    float lo = FLT_MAX, hi = -FLT_MAX;
    for (int i = threadIdx.x; i < n_partial; i += QT) { lo = fminf(lo, partial[2 * i]); hi = fmaxf(hi, partial[2 * i + 1]); }
    block_minmax(lo, hi, sh);
    if (blockIdx.x == 0 && threadIdx.x == 0) { lohi[0] = lo; lohi[1] = hi; }
    const float range = hi - lo;
    const bool flat = fabsf(range) < 1e-8f;
    auto enc = [&](float v) -> unsigned short {
        if (flat) return 0;
        const float u = (v - lo) / range;
        return (unsigned short)rintf(u * 65535.f);
    };
    const long long n4 = n >> 2;
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    ushort4 *q4 = reinterpret_cast<ushort4 *>(q);
    for (long long i = (long long)blockIdx.x * QT + threadIdx.x; i < n4; i += (long long)gridDim.x * QT) {
        const float4 v = x4[i];
        q4[i] = make_ushort4(enc(v.x), enc(v.y), enc(v.z), enc(v.w));
    }
    if (blockIdx.x == 0)
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += QT) q[i] = enc(x[i]);
}

__global__ __launch_bounds__(QT) void dequantize_kernel(const unsigned short *__restrict__ q, long long n, float lo,
                                                        float hi, float *__restrict__ x) {
    const float range = hi - lo;
    auto dec = [&](unsigned short v) { return ((float)v / 65535.f) * range + lo; };
    const long long n4 = n >> 2;
    const ushort4 *q4 = reinterpret_cast<const ushort4 *>(q);
    float4 *x4 = reinterpret_cast<float4 *>(x);
    for (long long i = (long long)blockIdx.x * QT + threadIdx.x; i < n4; i += (long long)gridDim.x * QT) {
        const ushort4 v = q4[i];
        x4[i] = make_float4(dec(v.x), dec(v.y), dec(v.z), dec(v.w));
    }
    if (blockIdx.x == 0)
        for (long long i = (n4 << 2) + threadIdx.x; i < n; i += QT) x[i] = dec(q[i]);
}

static int blocks_for(long long n) {
    const long long b = (n / 4 + QT - 1) / QT;
    return (int)(b < 1 ? 1 : (b > QB_MAX ? QB_MAX : b));
}

}  // namespace mftx

using namespace mftx;

extern "C" size_t mftx_quantize_workspace_bytes(void) { return (size_t)QB_MAX * 2 * sizeof(float); }

extern "C" int mftx_quantize_u16(const float *x, long long n, uint16_t *q, float *lohi, void *workspace,
                                 size_t workspace_bytes, void *stream) {
    if (!x || !q || !lohi || !workspace) return fail(MFTX_E_ARG, "quantize_u16: null pointer");
    if (n <= 0) return fail(MFTX_E_ARG, "quantize_u16: empty channel (numpy's amin raises on it too)");
    if (!aligned16(x) || (reinterpret_cast<uintptr_t>(q) & 7u)) return fail(MFTX_E_ALIGN, "quantize_u16: x must be 16-byte, q 8-byte aligned");
    if (workspace_bytes < mftx_quantize_workspace_bytes()) return fail(MFTX_E_WORKSPACE, "quantize_u16: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int nb = blocks_for(n);
    float *partial = (float *)workspace;
    ProfScope prof(PC_GLUE, s, 10.0 * (double)n);
    hipLaunchKernelGGL(minmax_partial_kernel, dim3(nb), dim3(QT), 0, s, x, n, partial);
    hipLaunchKernelGGL(quantize_kernel, dim3(nb), dim3(QT), 0, s, x, n, partial, nb, (unsigned short *)q, lohi);
    return check_launch("quantize_u16");
}

extern "C" int mftx_dequantize_u16(const uint16_t *q, long long n, float lo, float hi, float *x, void *stream) {
    if (!x || !q) return fail(MFTX_E_ARG, "dequantize_u16: null pointer");
    if (n <= 0) return fail(MFTX_E_ARG, "dequantize_u16: empty channel");
    if (!aligned16(x) || (reinterpret_cast<uintptr_t>(q) & 7u)) return fail(MFTX_E_ALIGN, "dequantize_u16: x must be 16-byte, q 8-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(PC_GLUE, s, 6.0 * (double)n);
    hipLaunchKernelGGL(dequantize_kernel, dim3(blocks_for(n)), dim3(QT), 0, s, (const unsigned short *)q, n, lo, hi, x);
    return check_launch("dequantize_u16");
}

// PNG scanline reconstruction (PNG spec 9.2, filter types 0-4), in place.  `rows` is the inflated
// IDAT stream of a non-interlaced image: height x (1 filter byte + row_bytes); the reconstructed
// pixels are compacted to the front (height x row_bytes).  Host code: byte-serial by definition
// (every byte depends on its left neighbour), so it lives here rather than in a Python loop.
// ---- shader copy ----------------------------------------------------------------------------------------------------------
// A pinned-host <-> device copy enqueued with hipMemcpyAsync goes through the SDMA queues, where an upload submitted
// while a download is pending waits for the compute that download waits for (profiles/r2_io_paths.txt: pinned uploads and
// downloads together serialise the loop).  Pinned host memory is mapped into the device's address space, so a KERNEL can
// move the bytes instead: it is ordered like any other kernel of its stream and touches no copy queue.
__global__ void copy_bytes_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, long long n16,
                                  const unsigned char *__restrict__ tail_src, unsigned char *__restrict__ tail_dst, int tail) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i];
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) tail_dst[threadIdx.x] = tail_src[threadIdx.x];
}

extern "C" int mftx_copy_bytes(const void *src, void *dst, long long n, void *stream) {
    if (!src || !dst || n < 0) return fail(MFTX_E_ARG, "copy_bytes: bad arguments");
    if (n == 0) return 0;
    if (!aligned16(src) || !aligned16(dst)) return fail(MFTX_E_ALIGN, "copy_bytes: both pointers must be 16-byte aligned");
    const long long n16 = n / 16;
    const long long blocks = (n16 + 255) / 256;
    hipLaunchKernelGGL(copy_bytes_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const uint4 *>(src), static_cast<uint4 *>(dst), n16,
                       static_cast<const unsigned char *>(src) + n16 * 16, static_cast<unsigned char *>(dst) + n16 * 16, (int)(n - n16 * 16));
    return check_launch("copy_bytes");
}

// ---- range guard of the split arithmetic (MFTX_ARITH_SPLIT): hi = fp16(x) needs |x| < 65504 ---------------------------------
__global__ void count_not_below_kernel(const float *__restrict__ x, long long n, float limit, unsigned *__restrict__ count) {
    unsigned bad = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        bad += !(fabsf(x[i]) < limit);                    // (NaN counts)
    for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor((int)bad, o);
    if ((threadIdx.x & 63) == 0 && bad) atomicAdd(count, bad);
}

extern "C" int mftx_count_not_below(const float *x, long long n, float limit, unsigned *count, void *stream) {
    if (!x || !count || n <= 0) return fail(MFTX_E_ARG, "count_not_below: bad arguments");
    const long long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(count_not_below_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, x, n, limit, count);
    return check_launch("count_not_below");
}

extern "C" int mftx_png_unfilter(uint8_t *rows, int height, int row_bytes, int bpp) {
    if (!rows || height <= 0 || row_bytes <= 0 || bpp <= 0 || bpp > 8) return fail(MFTX_E_ARG, "png_unfilter: bad arguments");
    const uint8_t *prev = nullptr;
    for (int y = 0; y < height; ++y) {
        const uint8_t *src = rows + (size_t)y * (row_bytes + 1);
        uint8_t *dst = rows + (size_t)y * row_bytes;     // dst <= src - 1 + ... : never overtakes the unread input
        const int ft = src[0];
        ++src;
        if (ft > 4) return fail(MFTX_E_ARG, "png_unfilter: bad filter type %d in row %d", ft, y);
        for (int i = 0; i < row_bytes; ++i) {
            const int a = i >= bpp ? dst[i - bpp] : 0;
            const int b = prev ? prev[i] : 0;
            const int c = (prev && i >= bpp) ? prev[i - bpp] : 0;
            int pred = 0;
            switch (ft) {
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: {
                    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                } break;
                default: break;
            }
            dst[i] = (uint8_t)(src[i] + pred);
        }
        prev = dst;
    }
    return 0;
}
