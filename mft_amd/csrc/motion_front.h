// Bodies of the two kernels that open every refinement iteration -- the correlation lookup
// (core/corr.py:30-51) and convf1 (core/update.py:147,154) -- as device functions, so that they can run
// as kernels of their own (csrc/corr.hip, per-op export and timing pass) and as one fused launch
// (csrc/raft_engine.hip).
#pragma once
#include "common.h"
#include <cstdlib>

namespace mftx {

typedef float f32x4lk __attribute__((ext_vector_type(4)));

constexpr int LK_WAVES = 4;
constexpr int LK_ROW = 20;          // floats per window row in LDS (10 or up to 16 used; 80 B keeps float4 stores aligned)
constexpr int LK_LVL = 10 * LK_ROW; // floats per level patch

struct LookupArgs {
    const float *lvl[4];
    const float *coords;
    float *out;
    int ld_out;
    int cells;      // P*h*w
    int n_per_img;  // h*w
    int hl[4], wl[4];
    int wb[2];              // block-grid width of levels 0, 1 (pyramid layout, common.h)
    unsigned hbwb[2];       // blocks per query of levels 0, 1
    long long stride[4];    // floats per query cell
    int ablate;     // tuning only (MFTX_LOOKUP_ABLATE): 1 no tap loads, 2 no stores, 3 neither
};

inline LookupArgs make_lookup_args(const float *const lvl[4], const float *coords, int P, int h, int w, float *out,
                                   int ld_out) {
    LookupArgs a;
    const PyramidLayout L = pyramid_layout(h, w);
    for (int l = 0; l < 4; ++l) { a.lvl[l] = lvl[l]; a.hl[l] = L.h[l]; a.wl[l] = L.w[l]; a.stride[l] = L.stride[l]; }
    for (int l = 0; l < 2; ++l) { a.wb[l] = L.wb[l]; a.hbwb[l] = (unsigned)(L.hb[l] * L.wb[l]); }
    a.coords = coords; a.out = out; a.ld_out = ld_out;
    a.cells = P * h * w; a.n_per_img = h * w;
    static const int ablate = tune_env("MFTX_LOOKUP_ABLATE", 0);
    a.ablate = ablate;
    return a;
}
inline double lookup_bytes(const LookupArgs &a) { return (double)a.cells * (4 * 100 * 4 + 8 + 324 * 4); }

// CPW cells per wave, all in flight together: the kernel is a chain of two memory round trips per
// cell (coordinates, then taps) with ~1.5 k issue cycles around them, and the chip holds 8192 waves
// for 28 672 cells -- the wave lifetime (5 us), not bandwidth, set the pace with one cell per wave.
//
// Levels 0 and 1 (8 x 4-float blocks, one 128-byte line each): the 10 x 10 window is fetched as 16-byte
// pieces -- 10 rows x 4 pieces starting at the 4-aligned column below the window, lanes 0..39, ONE
// buffer_load_dwordx4 per level; a piece never leaves its block row, rows / block columns outside the
// level get an out-of-range offset (the hardware returns zeros) and the cells of a piece beyond the
// level's width are zeroed when the piece is parked in LDS -- tap for tap grid_sample's zero padding.
// Levels 2 and 3 (row-major, <= 1 KiB per query at 512 x 512) keep the 2 dword taps per lane.
template <int CPW>
__device__ __forceinline__ void lookup_block_body(const LookupArgs &p, int block, int n_blocks) {
    // per wave and cell: 4 level patches [10][LK_ROW] + 4 x 4 bilinear weights
    __shared__ __attribute__((aligned(16))) float taps[LK_WAVES][CPW][4 * LK_LVL + 16];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;

    // lane-constant decode
    const int pr = lane >> 2, pc = lane & 3;                            // piece (row, column) of levels 0, 1 (lanes 0..39)
    const bool piece_lane = lane < 40;
    const int tr0 = lane / 10, tc0 = lane - tr0 * 10;                   // taps 0..63 of levels 2, 3
    const int tr1 = (lane + 64) / 10, tc1 = (lane + 64) - tr1 * 10;     // taps 64..99 (lanes 0..35)
    const bool t1_lane = lane < 36;
    int o_lvl[6], o_ab[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int o = lane + 64 * j;
        const int l = min(o / 81, 3);
        const int rem = o - l * 81;
        const int a = rem / 9, b = rem - a * 9;   // a offsets x, b offsets y
        o_lvl[j] = l;
        o_ab[j] = l * LK_LVL + b * LK_ROW + a;    // tap (row b, col a) of level l, before the level's column offset
    }
    // lanes 0..15 publish the bilinear weights: lane = 4 * level + {w00, w01, w10, w11}
    const int w_lvl = (lane >> 2) & 3, w_idx = lane & 3;

    const int groups = (p.cells + CPW - 1) / CPW;
    for (int g_v = block * LK_WAVES + wv; g_v < groups; g_v += n_blocks * LK_WAVES) {
        // CPW consecutive cells per wave: make that provable so the buffer descriptors stay in SGPRs
        const int cell0 = __builtin_amdgcn_readfirstlane(g_v) * CPW;
        f32x4lk pv[CPW][2];                 // levels 0, 1: this lane's piece
        float t0[CPW][2], t1[CPW][2];       // levels 2, 3: this lane's taps
        float my_fx[CPW], my_fy[CPW];
        int xo[CPW][2], xlim[CPW][2];       // window column inside the first piece; cells of my piece that are inside the level
        // Issue all tap loads of every cell back to back.  Each query's level slice is its own buffer; what
        // lies outside gets an out-of-range offset, which the hardware returns as 0 (= grid_sample's zero
        // padding) -- no branches, so the round trips overlap instead of serialising.
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
                    const_cast<float *>(p.lvl[l] + (long long)cell * p.stride[l]), 0, (unsigned)p.stride[l] * 4u, 0x00020000);
                if (l < 2) {
                    const int xa = x0 & ~3;                          // 4-aligned column at or below the window
                    xo[u][l] = x0 - xa;
                    const int xp = xa + 4 * pc;                      // first cell of my piece
                    const unsigned yy = (unsigned)(y0 + pr);
                    const unsigned bx = (unsigned)(xp >> 3);         // negative xp -> huge: out of range
                    const bool ok = piece_lane & (yy < H) & (xp >= 0) & (bx < (unsigned)p.wb[l]) & !(p.ablate & 1);
                    const unsigned off = (((yy >> 2) * (unsigned)p.wb[l] + bx) * 32u + (yy & 3u) * 8u + ((unsigned)xp & 7u)) * 4u;
                    pv[u][l] = __builtin_bit_cast(f32x4lk, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? off : 0x80000000u, 0, 0));
                    xlim[u][l] = (int)W - xp;                        // cells k < xlim of the piece are inside the level
                } else {
                    {   // unsigned compares fold the lower bounds in
                        const unsigned yy = (unsigned)(y0 + tr0), xx = (unsigned)(x0 + tc0);
                        const bool ok = (yy < H) & (xx < W) & !(p.ablate & 1);
                        t0[u][l - 2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                                      rs, ok ? (yy * W + xx) * 4u : 0x80000000u, 0, 0));
                    }
                    {
                        const unsigned yy = (unsigned)(y0 + tr1), xx = (unsigned)(x0 + tc1);
                        const bool ok = t1_lane & (yy < H) & (xx < W) & !(p.ablate & 1);
                        t1[u][l - 2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                                      rs, ok ? (yy * W + xx) * 4u : 0x80000000u, 0, 0));
                    }
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
                tp[4 * LK_LVL + lane] = ax * ay;
            }
#pragma unroll
            for (int l = 0; l < 2; ++l) {
                if (piece_lane) {
                    f32x4lk v = pv[u][l];
                    const int lim = xlim[u][l];
                    v.x = lim > 0 ? v.x : 0.f; v.y = lim > 1 ? v.y : 0.f; v.z = lim > 2 ? v.z : 0.f; v.w = lim > 3 ? v.w : 0.f;
                    *reinterpret_cast<f32x4lk *>(tp + l * LK_LVL + pr * LK_ROW + pc * 4) = v;
                }
            }
#pragma unroll
            for (int l = 2; l < 4; ++l) {
                tp[l * LK_LVL + tr0 * LK_ROW + tc0] = t0[u][l - 2];
                if (t1_lane) tp[l * LK_LVL + tr1 * LK_ROW + tc1] = t1[u][l - 2];
            }
            // LDS operations of one wave complete in issue order, so the wave can
            // read back what its other lanes just wrote without a barrier.
            float *dst = p.out + (long long)(cell0 + u) * p.ld_out;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int o = lane + 64 * j;
                if ((j < 5 || o < 324) && (!(p.ablate & 2) || (j == 0 && lane == 0))) {
                    const int l = o_lvl[j];
                    const float4 wq = *reinterpret_cast<const float4 *>(tp + 4 * LK_LVL + l * 4);
                    const float *t4 = tp + o_ab[j] + (l == 0 ? xo[u][0] : l == 1 ? xo[u][1] : 0);
                    const float v00 = t4[0], v01 = t4[1], v10 = t4[LK_ROW], v11 = t4[LK_ROW + 1];
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
    int out_split;      // flo1 and the flow tail of hx are written in split form (common.h)
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
        const bool st = x < w && live;                       // (uniform per block)
        const long long cell = st ? img_base + (long long)y * w + x : 0;
        if (q.out_split) {                                   // every lane takes part in the pair exchange
            store_split_pairwise(flo1 + cell * 128, co, relu_keep_nan(acc[t]), st);
            store_split_pairwise(hx + cell * 384, 382 + (co & 1), patch[3][2 * (t + 3) + (co & 1)], st && co < 2);
        } else if (st) {
            flo1[cell * 128 + co] = relu_keep_nan(acc[t]);
            if (co < 2) hx[cell * 384 + 382 + co] = patch[3][2 * (t + 3) + co];
        }
    }
}

}  // namespace mftx
