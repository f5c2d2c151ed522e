// Correlation pyramid pooling and the multi-scale 9x9 lookup (HBM-bound).
#include "common.h"
#include "profile.h"
#include "motion_front.h"
#include <cstdlib>

namespace mftx {

// ---------------------------------------------------------------------------
// Pyramid: levels 1..3 from level 0, one workgroup per query row.
// core/corr.py:26-28: 3x avg_pool2d(2, stride 2) over the target dims (floor
// sizes).  Sum order ((v00 + v01) + v10) + v11, then * 0.25 -- the order ATen's
// avg_pool2d uses -- so the result is bit-identical to the CPU reference.
// The row of level 0 is read once from HBM; levels 1 and 2 stay in LDS while
// the next level is formed.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void corr_pool_kernel(const float *__restrict__ lvl0, int h, int w,
                                                        float *__restrict__ lvl1, float *__restrict__ lvl2,
                                                        float *__restrict__ lvl3) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int h1 = h >> 1, w1 = w >> 1, h2 = h >> 2, w2 = w >> 2, h3 = h >> 3, w3 = w >> 3;
    float *s1 = sm;                 // h1*w1
    float *s2 = sm + h1 * w1;       // h2*w2
    const long long row = blockIdx.x;
    const float *src = lvl0 + row * (long long)h * w;
    float *d1 = lvl1 + row * (long long)h1 * w1;
    float *d2 = lvl2 + row * (long long)h2 * w2;
    float *d3 = lvl3 + row * (long long)h3 * w3;
    // four outputs per thread and trip: all eight 8-byte loads first, then the stores (a store between
    // two loads makes the compiler drain vmcnt before the second load's data can be used)
    for (int i0 = threadIdx.x; i0 < h1 * w1; i0 += 4 * blockDim.x) {
        float2 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * blockDim.x;
            a[u] = b[u] = make_float2(0.f, 0.f);
            if (i < h1 * w1) {
                const int y = i / w1, x = i - y * w1;
                const float *p = src + (2 * y) * w + 2 * x;       // even offset: 8-byte aligned when w is even
                if ((w & 1) == 0) {
                    a[u] = *reinterpret_cast<const float2 *>(p);
                    b[u] = *reinterpret_cast<const float2 *>(p + w);
                } else {
                    a[u] = make_float2(p[0], p[1]);
                    b[u] = make_float2(p[w], p[w + 1]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * blockDim.x;
            if (i < h1 * w1) {
                const float v = (((a[u].x + a[u].y) + b[u].x) + b[u].y) * 0.25f;
                s1[i] = v;
                d1[i] = v;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < h2 * w2; i += blockDim.x) {
        const int y = i / w2, x = i - y * w2;
        const float *p = s1 + (2 * y) * w1 + 2 * x;
        const float v = (((p[0] + p[1]) + p[w1]) + p[w1 + 1]) * 0.25f;
        s2[i] = v;
        d2[i] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < h3 * w3; i += blockDim.x) {
        const int y = i / w3, x = i - y * w3;
        const float *p = s2 + (2 * y) * w2 + 2 * x;
        d3[i] = (((p[0] + p[1]) + p[w2]) + p[w2 + 1]) * 0.25f;
    }
}

int launch_corr_pool(const float *lvl0, int rows, int h, int w, float *lvl1, float *lvl2, float *lvl3,
                     hipStream_t s) {
    const size_t lds = ((size_t)(h >> 1) * (w >> 1) + (size_t)(h >> 2) * (w >> 2)) * sizeof(float);
    if (lds > 150 * 1024) return fail(MFTX_E_ARG, "corr_pool: feature map too large for the LDS tile");
    static size_t attr = 0;
    if (lds > attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(corr_pool_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr = lds;
    }
    ProfScope prof(PC_CORR_POOL, s, 4.0 * rows * ((double)h * w + (h >> 1) * (w >> 1) + (h >> 2) * (w >> 2) + (h >> 3) * (w >> 3)));
    hipLaunchKernelGGL(corr_pool_kernel, dim3(rows), dim3(256), lds, s, lvl0, h, w, lvl1, lvl2, lvl3);
    return check_launch("corr_pool");
}

// ---------------------------------------------------------------------------
// Lookup (core/corr.py:30-51, core/utils/utils.py:98-112).
//
// One wave per query cell.  For each of the 4 levels the 81 window samples
// share one fractional offset, so the cell needs only the 10x10 integer taps
// around floor(c / 2^l) - 4: the wave pulls those 400 taps (zero outside the
// level, matching grid_sample's zero padding tap by tap) into its private LDS
// slab, then every lane blends 4 neighbours for its output channels and writes
// them pixel-major, 256 contiguous bytes per store.  Algorithmic traffic per
// cell: 4*100*4 B read + 8 B coords + 324*4 B written.
//
// Coordinates: the reference normalises to [-1,1] and grid_sample maps back;
// that round trip moves a coordinate by O(1e-6) px, and zero-padded bilinear
// sampling is continuous in the coordinate, so sampling directly at
// c/2^l + (a-4) agrees to O(1e-6 * |grad V|).
// ---------------------------------------------------------------------------
int launch_corr_lookup(const float *const lvl[4], const float *coords, int P, int h, int w, float *out,
                       int ld_out, hipStream_t s) {
    const LookupArgs a = make_lookup_args(lvl, coords, P, h, w, out, ld_out);
    static const int cpw = [] { const char *e = getenv("MFTX_LOOKUP_CPW"); return e ? atoi(e) : 2; }();
    // 4 levels x 10x10 unique taps read + coords + 324 outputs written, per cell
    ProfScope prof(PC_LOOKUP, s, lookup_bytes(a));
    if (cpw == 1)
        hipLaunchKernelGGL(corr_lookup_kernel<1>, dim3(cdiv(a.cells, LK_WAVES)), dim3(64 * LK_WAVES), 0, s, a);
    else if (cpw == 4)
        hipLaunchKernelGGL(corr_lookup_kernel<4>, dim3(cdiv(cdiv(a.cells, 4), LK_WAVES)), dim3(64 * LK_WAVES), 0, s, a);
    else
        hipLaunchKernelGGL(corr_lookup_kernel<2>, dim3(cdiv(cdiv(a.cells, 2), LK_WAVES)), dim3(64 * LK_WAVES), 0, s, a);
    return check_launch("corr_lookup");
}

}  // namespace mftx
