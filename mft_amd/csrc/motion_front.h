// Bodies of the two kernels that open every refinement iteration -- the correlation lookup
// (core/corr.py:30-51) and convf1 (core/update.py:147,154) -- as device functions, so that they can run
// as kernels of their own (csrc/corr.hip, per-op export and timing pass) and as one fused launch
// (csrc/raft_engine.hip).
#pragma once
#include "common.h"
#include <cstdlib>

namespace mftx {

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

inline LookupArgs make_lookup_args(const float *const lvl[4], const float *coords, int P, int h, int w, float *out,
                                   int ld_out) {
    LookupArgs a;
    for (int l = 0; l < 4; ++l) { a.lvl[l] = lvl[l]; a.hl[l] = h >> l; a.wl[l] = w >> l; }
    a.coords = coords; a.out = out; a.ld_out = ld_out;
    a.cells = P * h * w; a.n_per_img = h * w;
    static const int ablate = [] { const char *e = getenv("MFTX_LOOKUP_ABLATE"); return e ? atoi(e) : 0; }();
    a.ablate = ablate;
    return a;
}
inline double lookup_bytes(const LookupArgs &a) { return (double)a.cells * (4 * 100 * 4 + 8 + 324 * 4); }

// CPW cells per wave, all in flight together: the kernel is a chain of two memory round trips per
// cell (coordinates, then taps) with ~1.5 k issue cycles around them, and the chip holds 8192 waves
// for 28 672 cells -- the wave lifetime (5 us), not bandwidth, set the pace with one cell per wave.
template <int CPW>
__device__ __forceinline__ void lookup_block_body(const LookupArgs &p, int block, int n_blocks) {
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
    for (int g_v = block * LK_WAVES + wv; g_v < groups; g_v += n_blocks * LK_WAVES) {
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

template <int CPW>
__global__ __launch_bounds__(64 * LK_WAVES) void corr_lookup_kernel(LookupArgs p) {
    lookup_block_body<CPW>(p, blockIdx.x, gridDim.x);
}

// convf1: 7x7 conv over the 2-channel flow (= coords1 - grid), 2 -> 128, ReLU
// (core/update.py:147,154).  K = 98 is too thin for MFMA: direct VALU kernel,
// one thread per output channel, 16 cells of one row per block, flow patch in LDS.
// Also drops flow into channels 382..383 of hx (motion_features' tail,
// core/update.py:160).
constexpr int F1_CELLS = 16;
struct ConvF1Args {
    const float *coords1, *w98 /* [98][128] */, *bias;
    float *flo1, *hx;
    int h, w, strips_per_row, n_strips;
};
// one strip of 16 cells by 128 threads (`tid` 0..127); `patch` is that half-block's LDS slab.  Every
// thread of the block must call this the same number of times (it synchronises the block).
__device__ __forceinline__ void convf1_strip(const ConvF1Args &q, int strip_id, float (*patch)[48], int tid) {
    const float *__restrict__ coords1 = q.coords1;
    const float *__restrict__ w98 = q.w98;
    const float *__restrict__ bias = q.bias;
    float *__restrict__ flo1 = q.flo1;
    float *__restrict__ hx = q.hx;
    const int h = q.h, w = q.w, strips_per_row = q.strips_per_row;
    const bool live = strip_id < q.n_strips;
    if (!live) strip_id = 0;                         // (keeps the barrier count uniform; nothing is stored)
    // flow patch of the strip: 7 rows x (16 + 6) cells x (fx, fy), rows padded to 48 floats
    const int strip = strip_id % strips_per_row;
    const int rowid = strip_id / strips_per_row;     // img*h + y
    const int y = rowid % h;
    const long long img_base = (long long)(rowid / h) * h * w;
    const int x0 = strip * F1_CELLS;
    for (int i = tid; i < 7 * (F1_CELLS + 6); i += 128) {
        const int r = i / (F1_CELLS + 6), c = i - r * (F1_CELLS + 6);
        const int yy = y + r - 3, xx = x0 + c - 3;
        float fx = 0.f, fy = 0.f;
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) {
            const long long cell = img_base + (long long)yy * w + xx;
            fx = coords1[2 * cell] - (float)xx;
            fy = coords1[2 * cell + 1] - (float)yy;
        }
        patch[r][2 * c] = fx;
        patch[r][2 * c + 1] = fy;
    }
    __syncthreads();
    const int co = tid;
    float acc[F1_CELLS];
    const float b = bias[co];
#pragma unroll
    for (int t = 0; t < F1_CELLS; ++t) acc[t] = b;
    // this output channel's 14 weights of filter row ky are loaded one row ahead (register double
    // buffer): taken inside the row loop their L2 latency showed 7 times per strip
    float wc[14], wn[14];
#pragma unroll
    for (int q = 0; q < 14; ++q) wc[q] = w98[q * 128 + co];
#pragma unroll 1
    for (int ky = 0; ky < 7; ++ky) {
        const int kyn = ky < 6 ? ky + 1 : 6;
#pragma unroll
        for (int q = 0; q < 14; ++q) wn[q] = w98[(kyn * 14 + q) * 128 + co];
        // the whole patch row goes to registers once (11 broadcast ds_read_b128), then 7 x 16 x 2 FMAs
        float row[44];
#pragma unroll
        for (int q = 0; q < 11; ++q) {
            const float4 v = *reinterpret_cast<const float4 *>(&patch[ky][4 * q]);
            row[4 * q] = v.x; row[4 * q + 1] = v.y; row[4 * q + 2] = v.z; row[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
            const float w0 = wc[2 * kx], w1 = wc[2 * kx + 1];
#pragma unroll
            for (int t = 0; t < F1_CELLS; ++t)
                acc[t] += w0 * row[2 * (t + kx)] + w1 * row[2 * (t + kx) + 1];
        }
#pragma unroll
        for (int q = 0; q < 14; ++q) wc[q] = wn[q];
    }
#pragma unroll
    for (int t = 0; t < F1_CELLS; ++t) {
        const int x = x0 + t;
        if (x < w && live) {
            const long long cell = img_base + (long long)y * w + x;
            flo1[cell * 128 + co] = fmaxf(acc[t], 0.f);
            if (co < 2) hx[cell * 384 + 382 + co] = patch[3][2 * (t + 3) + co];
        }
    }
}

}  // namespace mftx
