// Correlation pyramid pooling and the multi-scale 9x9 lookup (HBM-bound).
#include "common.h"
#include "profile.h"
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
constexpr int LK_WAVES = 4;

struct LookupArgs {
    const float *lvl[4];
    const float *coords;
    float *out;
    int ld_out;
    int cells;      // P*h*w
    int n_per_img;  // h*w
    int hl[4], wl[4];
    int ablate;     // tuning only (MFTX_LOOKUP_ABLATE): 1 no tap loads, 2 no stores, 3 neither
};

// CPW cells per wave, all in flight together: the kernel is a chain of two memory round trips per
// cell (coordinates, then taps) with ~1.5 k issue cycles around them, and the chip holds 8192 waves
// for 28 672 cells -- the wave lifetime (5 us), not bandwidth, set the pace with one cell per wave.
template <int CPW>
__global__ __launch_bounds__(64 * LK_WAVES) void corr_lookup_kernel(LookupArgs p) {
    // per wave and cell: 4 levels x 128 tap slots (100 used) + 4 x 4 bilinear weights
    __shared__ __attribute__((aligned(16))) float taps[LK_WAVES][CPW][4 * 128 + 16];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;

    // lane-constant decode: tap slots (2 per level per lane) and output slots (6 per lane)
    const int tr0 = lane / 10, tc0 = lane - tr0 * 10;                   // taps 0..63
    const int tr1 = (lane + 64) / 10, tc1 = (lane + 64) - tr1 * 10;     // taps 64..99 (lanes 0..35)
    const bool t1_lane = lane < 36;
    int o_off[6], o_w[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int o = lane + 64 * j;
        const int l = min(o / 81, 3);
        const int rem = o - l * 81;
        const int a = rem / 9, b = rem - a * 9;   // a offsets x, b offsets y
        o_off[j] = l * 128 + b * 10 + a;          // tap (row b, col a) of level l
        o_w[j] = 512 + l * 4;                     // that level's 4 weights
    }
    // lanes 0..15 publish the bilinear weights: lane = 4 * level + {w00, w01, w10, w11}
    const int w_lvl = (lane >> 2) & 3, w_idx = lane & 3;

    const int groups = (p.cells + CPW - 1) / CPW;
    for (int g_v = blockIdx.x * LK_WAVES + wv; g_v < groups; g_v += gridDim.x * LK_WAVES) {
        // CPW consecutive cells per wave: make that provable so the buffer descriptors stay in SGPRs
        const int cell0 = __builtin_amdgcn_readfirstlane(g_v) * CPW;
        float t0[CPW][4], t1[CPW][4];
        float my_fx[CPW], my_fy[CPW];
        // Issue all 8 tap loads of every cell back to back.  Each level slice is its
        // own buffer; taps outside the slice get an out-of-range offset, which
        // the hardware returns as 0 (= grid_sample's zero padding) -- no branches,
        // so the round trips overlap instead of serialising.
#pragma unroll
        for (int u = 0; u < CPW; ++u) {
            const int cell = min(cell0 + u, p.cells - 1);            // (odd tail: recomputed, not stored)
            const float2 c = reinterpret_cast<const float2 *>(p.coords)[cell];
            my_fx[u] = my_fy[u] = 0.f;
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                const float sx = c.x / (float)(1 << l), sy = c.y / (float)(1 << l);
                const float flx = floorf(sx), fly = floorf(sy);
                if (w_lvl == l) { my_fx[u] = sx - flx; my_fy[u] = sy - fly; }
                // clamp so that the int conversion is defined for wild coordinates
                const int x0 = (int)fminf(fmaxf(flx, -1.0e6f), 1.0e6f) - 4;
                const int y0 = (int)fminf(fmaxf(fly, -1.0e6f), 1.0e6f) - 4;
                const unsigned H = p.hl[l], W = p.wl[l];
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float *>(p.lvl[l] + (long long)cell * H * W), 0, H * W * 4u, 0x00020000);
                {   // unsigned compares fold the lower bounds in
                    const unsigned yy = (unsigned)(y0 + tr0), xx = (unsigned)(x0 + tc0);
                    const bool ok = (yy < H) & (xx < W) & !(p.ablate & 1);
                    t0[u][l] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                             rs, ok ? (yy * W + xx) * 4u : 0x80000000u, 0, 0));
                }
                {
                    const unsigned yy = (unsigned)(y0 + tr1), xx = (unsigned)(x0 + tc1);
                    const bool ok = t1_lane & (yy < H) & (xx < W) & !(p.ablate & 1);
                    t1[u][l] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                             rs, ok ? (yy * W + xx) * 4u : 0x80000000u, 0, 0));
                }
            }
        }
#pragma unroll
        for (int u = 0; u < CPW; ++u) {
            if (cell0 + u >= p.cells) break;
            float *tp = taps[wv][u];
            if (lane < 16) {
                const float ax = (w_idx & 1) ? my_fx[u] : 1.f - my_fx[u];
                const float ay = (w_idx & 2) ? my_fy[u] : 1.f - my_fy[u];
                tp[512 + lane] = ax * ay;
            }
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                tp[l * 128 + lane] = t0[u][l];
                tp[l * 128 + 64 + lane] = t1[u][l];       // lanes >= 36 park zeros in the padding
            }
            // LDS operations of one wave complete in issue order, so the wave can
            // read back what its other lanes just wrote without a barrier.
            float *dst = p.out + (long long)(cell0 + u) * p.ld_out;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int o = lane + 64 * j;
                if ((j < 5 || o < 324) && (!(p.ablate & 2) || (j == 0 && lane == 0))) {
                    const float4 wq = *reinterpret_cast<const float4 *>(tp + o_w[j]);
                    const float *t4 = tp + o_off[j];
                    const float v00 = t4[0], v01 = t4[1], v10 = t4[10], v11 = t4[11];
                    dst[o] = v00 * wq.x + v01 * wq.y + v10 * wq.z + v11 * wq.w;
                }
            }
        }
    }
}

int launch_corr_lookup(const float *const lvl[4], const float *coords, int P, int h, int w, float *out,
                       int ld_out, hipStream_t s) {
    LookupArgs a;
    for (int l = 0; l < 4; ++l) { a.lvl[l] = lvl[l]; a.hl[l] = h >> l; a.wl[l] = w >> l; }
    a.coords = coords; a.out = out; a.ld_out = ld_out;
    a.cells = P * h * w; a.n_per_img = h * w;
    static const int ablate = [] { const char *e = getenv("MFTX_LOOKUP_ABLATE"); return e ? atoi(e) : 0; }();
    a.ablate = ablate;
    static const int cpw = [] { const char *e = getenv("MFTX_LOOKUP_CPW"); return e ? atoi(e) : 2; }();
    // 4 levels x 10x10 unique taps read + coords + 324 outputs written, per cell
    ProfScope prof(PC_LOOKUP, s, (double)a.cells * (4 * 100 * 4 + 8 + 324 * 4));
    if (cpw == 1)
        hipLaunchKernelGGL(corr_lookup_kernel<1>, dim3(cdiv(a.cells, LK_WAVES)), dim3(64 * LK_WAVES), 0, s, a);
    else if (cpw == 4)
        hipLaunchKernelGGL(corr_lookup_kernel<4>, dim3(cdiv(cdiv(a.cells, 4), LK_WAVES)), dim3(64 * LK_WAVES), 0, s, a);
    else
        hipLaunchKernelGGL(corr_lookup_kernel<2>, dim3(cdiv(cdiv(a.cells, 2), LK_WAVES)), dim3(64 * LK_WAVES), 0, s, a);
    return check_launch("corr_lookup");
}

}  // namespace mftx
