// Shared host-side helpers for libmftx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include "../../include/mftx.h"

namespace mftx {

void set_error(const char *fmt, ...);

inline int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    set_error("%s", buf);
    return code;
}

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

// Tuning switches of tools/ (micro-benchmarks, A/B builds): environment variables are read in -DMFTX_TUNING builds only;
// the product build compiles the defaults in and reads no global state.
inline int tune_env(const char *name, int dflt) {
#ifdef MFTX_TUNING
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return cdiv(a, b) * b; }

// ---- correlation pyramid layout (per query cell) ------------------------------------
// Levels 0 and 1 are stored in BLOCKS of 8 x 4 floats (x fastest; one 128-byte line each): the 10 x 10
// lookup window then touches (1 + 9/8)(1 + 9/4) = 6.9 lines on average instead of 12.8 (row-major 64-wide
// rows) / 10 (level 1).  Element (y, x) of level l < 2 lives at ((y >> 2) * wb[l] + (x >> 3)) * 32 +
// (y & 3) * 8 + (x & 7).  The block grid is padded so that level 0 is made of whole SUPER-BLOCKS (2 x 2
// blocks = 16 x 8 cells): one super-block is one N tile of the volume GEMM, whose epilogue also forms the
// level-1 block and the 4 x 2 / 2 x 1 pieces of levels 2 / 3 it covers.  Levels 2 and 3 (<= 1 KiB per query
// at 512 x 512) stay row-major [h_l][w_l]; every per-query stride is a multiple of 4 floats.
struct PyramidLayout {
    int h[4], w[4];          // level sizes (floor halving, core/corr.py:26-28)
    int hb[2], wb[2];        // block grid of levels 0, 1
    int sbh, sbw;            // super-block grid (= block grid of level 1)
    long long stride[4];     // floats per query cell
};
inline PyramidLayout pyramid_layout(int h, int w) {
    PyramidLayout L;
    for (int l = 0; l < 4; ++l) { L.h[l] = h >> l; L.w[l] = w >> l; }
    L.sbh = (h + 7) / 8; L.sbw = (w + 15) / 16;
    L.hb[0] = 2 * L.sbh; L.wb[0] = 2 * L.sbw; L.hb[1] = L.sbh; L.wb[1] = L.sbw;
    L.stride[0] = (long long)L.hb[0] * L.wb[0] * 32;
    L.stride[1] = (long long)L.hb[1] * L.wb[1] * 32;
    L.stride[2] = ((long long)L.h[2] * L.w[2] + 3) / 4 * 4;
    L.stride[3] = ((long long)L.h[3] * L.w[3] + 3) / 4 * 4;
    return L;
}

// ---- kernel launchers shared between the per-op exports and the RAFT engine
// SPLIT storage of an fp32 tensor row (the A operands of the split-arithmetic GEMMs, written that way by their
// producers): every 8 consecutive channels are 32 bytes [hi x 8 | lo x 8] of fp16, hi = fp16(x), lo = fp16((x - hi) * 2048)
// -- same size, same row stride, channel c at byte split_row_offset(c) (its low half 16 bytes further).
__host__ __device__ inline int split_row_offset(int c) { return (c >> 3) * 32 + (c & 7) * 2; }
#ifdef __HIPCC__
// (hi, lo) halves of a value, bit for bit what the GEMM's register split produces (round to nearest twice)
__device__ __forceinline__ unsigned split_halves(float v) {
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)((v - (float)h) * 2048.f);
    return (unsigned)__builtin_bit_cast(unsigned short, h) | ((unsigned)__builtin_bit_cast(unsigned short, l) << 16);
}
// Channel c of a split-form row, one value per lane, the lanes of a pair (lane ^ 1) holding channels c and c ^ 1 with
// the parity of c = the parity of the lane: the even lane stores the pair's high halves, the odd lane the low halves --
// one 4-byte store per lane, like the fp32 form.  EVERY lane of the wave must call this (the exchange is a shuffle);
// `store` says whether this lane's pair is written.
__device__ __forceinline__ void store_split_pairwise(float *row, int c, float v, bool store) {
    const unsigned mine = split_halves(v);
    const unsigned other = (unsigned)__shfl_xor((int)mine, 1);
    const bool odd = (c & 1) != 0;
    const unsigned word = odd ? ((other >> 16) | (mine & 0xffff0000u)) : ((mine & 0xffffu) | (other << 16));
    char *dst = reinterpret_cast<char *>(row) + split_row_offset(c & ~1) + (odd ? 16 : 0);
    if (store) *reinterpret_cast<unsigned *>(dst) = word;
}
// four consecutive channels c .. c + 3 (c % 4 == 0) of a split-form row <-> four floats
__device__ __forceinline__ float4 load_split4(const float *row, int c) {
    const char *src = reinterpret_cast<const char *>(row) + split_row_offset(c);
    const uint2 h = *reinterpret_cast<const uint2 *>(src), l = *reinterpret_cast<const uint2 *>(src + 16);
    auto dec = [](unsigned hh, unsigned ll, int hi) {
        const _Float16 a = __builtin_bit_cast(_Float16, (unsigned short)(hi ? hh >> 16 : hh));
        const _Float16 b = __builtin_bit_cast(_Float16, (unsigned short)(hi ? ll >> 16 : ll));
        return (float)a + (float)b * (1.f / 2048.f);
    };
    return make_float4(dec(h.x, l.x, 0), dec(h.x, l.x, 1), dec(h.y, l.y, 0), dec(h.y, l.y, 1));
}
__device__ __forceinline__ void store_split4v(float *row, int c, float4 v) {
    const unsigned a = split_halves(v.x), b = split_halves(v.y), cc = split_halves(v.z), d = split_halves(v.w);
    char *dst = reinterpret_cast<char *>(row) + split_row_offset(c);
    *reinterpret_cast<uint2 *>(dst) = make_uint2((a & 0xffffu) | (b << 16), (cc & 0xffffu) | (d << 16));
    *reinterpret_cast<uint2 *>(dst + 16) = make_uint2((a >> 16) | (b & 0xffff0000u), (cc >> 16) | (d & 0xffff0000u));
}
#endif
// ReLU that keeps a NaN (fmaxf(NaN, 0) is 0): an operand outside the split arithmetic's fp16 range turns into NaN
// products, and that NaN must reach the outputs instead of being clipped to a finite wrong value by the next activation
__host__ __device__ inline float relu_keep_nan(float v) { return v < 0.f ? 0.f : v; }
int launch_conv(const mftx_conv_desc &d, hipStream_t s, int tile = -1);      // tile >= 0: forced tile shape (mftx_conv2d_tile)
int launch_conv_pair(const mftx_conv_desc &a, const mftx_conv_desc &b, hipStream_t s);   // two independent ReLU convs, one launch
// conv whose epilogue is a GRU gate (see conv_gemm.hip)
struct GruEpilogue {
    int mode;          // 1: z|r gates  2: candidate + blend
    float *hx;         // [M][ld_hx], hidden state in channels [0,128)
    int ld_hx;
    float *z;          // [M][128]
    float *rh;         // [M][128]
    // split arithmetic with split-form activations (mftx_conv_desc.out_split): hx and rh are written in split form, and
    // the gate algebra reads / writes this fp32 copy of h instead ([M][ld_hf]); null: h lives in hx as fp32
    float *hf = nullptr;
    int ld_hf = 0;
};
int launch_conv_gru(const mftx_conv_desc &d, const GruEpilogue &g, hipStream_t s);
int launch_split_weights(const float *wpk, void *out, long long n_floats, hipStream_t s);
bool conv_small_applicable(const mftx_conv_desc &d);
int launch_conv_small(const mftx_conv_desc &d, hipStream_t s, float *accum = nullptr, int ld_accum = 0);
// all-pairs correlation volume + its 3 pooled levels in one launch (pyramid layout above)
int launch_corr_pyramid(const float *f1, const float *f2, int P, int C, int h, int w, float *const lvl[4], hipStream_t s,
                        float *f2_split = nullptr, int tile_resident = 1);
// the split-arithmetic volume with the target super-block resident in LDS (csrc/volume_tile.hip); f2s: split form of f2
bool volume_tile_applicable(int C);
// per-pair feature pointers (mftx_raft_refine_gather: the pairs' maps need not form one tensor)
constexpr int MFTX_MAX_GATHER = 16;
struct PairPtrs { const float *p[MFTX_MAX_GATHER]; };
// f1p (optional): pair bz's query features at f1p->p[bz] instead of f1 + bz * N * 256; f2_bstride: floats between the pairs'
// split target maps in f2s (0: every pair correlates against the same map)
int launch_volume_tile(const float *f1, const float *f2s, int P, int h, int w, float *const lvl[4], hipStream_t s,
                       const PairPtrs *f1p = nullptr, long long f2_bstride = -1);
int launch_corr_lookup(const float *const lvl[4], const float *coords, int P, int h, int w,
                       float *out, int ld_out, hipStream_t s);
// lookup fused into convc1 (csrc/lookup_convc1.hip): out = relu(convc1(lookup(coords)) + bias), [M][ld_out], fp32 or split form
int launch_pack_lookup_convc1(const float *w, int ld_w, void *out, hipStream_t s);
bool lookup_convc1_applicable(int P, int h, int w, int ld_out);
int launch_lookup_convc1(const float *const lvl[4], const float *coords, int P, int h, int w, const void *wf,
                         const float *bias, float *out, int ld_out, int out_split, hipStream_t s);
// the motion encoder's flow branch as one kernel (csrc/flow_branch.hip): out = relu(convf2(relu(convf1(coords - grid)))), split form
int launch_pack_flow_branch(const float *w98, const float *w2pk, void *out, hipStream_t s);
int launch_flow_branch(const float *coords, int P, int h, int w, const void *wf, const float *b1, const float *b2, float *out,
                       int ld_out, float *hx, int ld_hx, hipStream_t s, const float *T = nullptr, const float *b2h = nullptr,
                       float *coords_out = nullptr, float *delta_out = nullptr);
// tile-resident convolution GEMM (csrc/tile_conv.hip): the update block's layers whose input tile fits a CU's LDS
struct TileConvLaunch {
    const float *a0; int lda0;          // 128 channels per cell in split form
    const float *a1; int lda1;          // 128 more (cin = 256) or null
    int cin;                            // 128 or 256
    const void *wf;                     // launch_pack_tile_conv
    const float *bias;                  // [N] or null
    const float *addend; int ld_addend; // pre-activation addend [M][N] or null
    float *out; int ldo; int out_split; // epi 0 (linear) / 1 (relu)
    float *z, *rh, *hf, *hx; int ld_hf, ld_hx;      // epi 2 (z | r gates) / 3 (candidate + blend): as GruEpilogue
    const void *wproj; float *tout;     // epi 4 (flow head: relu, then the next layer's 3 x 3 x 2 filter as [256 x 18] partial products -> tout [M][18])
    int P, h, w, N, kh, kw, epi;
    int cells;                          // cells per tile: 128, 64, 32; 0: by how the tiles fill the chip (same bits either way)
};
// one SepConvGRU pass as one kernel (tile_conv.hip: gru_half_kernel); pass 0: 1 x 5, pass 1: 5 x 1
struct GruHalfLaunch {
    const float *h_in; int ld_hin; const float *mo; int ld_mo;      // h and the motion features, split form, 128 channels each
    const void *wzr, *wq;                                           // launch_pack_tile_conv streams (N = 256 / 128, cin = 256)
    const float *pre_zr, *pre_q;                                    // context parts + bias [M][256] / [M][128]
    float *z; const float *hf_in; float *hf_out;                    // z scratch [M][128]; h in fp32 [M][128]: read here, written there
    float *h_out; int ld_hout;                                      // new h in split form (a different buffer than h_in)
    int P, h, w, pass;
    int cells;                                                      // R cells per tile: 128, 64, 32; 0: by how the tiles fill the chip
};
int launch_gru_half(const GruHalfLaunch &d, hipStream_t s);
int launch_pack_flow_head(const float *w2pk, void *out, hipStream_t s);
// 3 x 3 over 256 split-form channels, tile-resident in two channel passes (tile_conv.hip: tile_conv2p_kernel): convc2 (N = 192), conv (126 of 128)
int launch_pack_tile_conv2p(const float *wpk, int N, int cin_pad, void *out, hipStream_t s);
int launch_tile_conv2p(const float *a, int lda, const void *wf, const float *bias, float *out, int ldo, int n_valid, int P, int h, int w, int cells,
                       hipStream_t s);
// the occlusion + uncertainty heads as one tile-resident kernel + a stencil sum (tile_conv.hip: ou_head_kernel)
constexpr size_t OU_HEAD_WTILE_BYTES = 8ull * 5 * 9 * 9 * 128 * 16, OU_HEAD_WPROJ_BYTES = 32768;
int launch_pack_ou_head(const float *w1pk, int cin_pad, const float *w2pk, void *wtile, void *wproj, hipStream_t s);
struct OuGather { const float *hx, *corr; int ld_corr; const float *coords1, *delta; float *flow_lr; };     // the parts of the heads' input (hx: [M][384] split form)
int launch_ou_heads(const float *a, int lda, int P, int h, int w, const void *wtile, const float *b1, const void *wproj, const float *b2, float *T,
                    float *out, int ld_out, int cells, hipStream_t s, const OuGather *ga = nullptr);
int launch_flow_head_sum(const float *T, const float *b2, float *delta, const float *coords_in, float *coords_out, int P, int h, int w, hipStream_t s);
bool tile_conv_applicable(int kh, int kw, int cin, int N);
bool tile_conv_small_tiles_fill(int P, int h, int w);               // at least half a round of the chip in 32-cell tiles
bool tile_conv_fills_chip(int P, int h, int w, int kh, int kw);     // its tiles come in rounds of the chip that are >= 5/8 full
int launch_pack_tile_conv(const float *wpk, int N, int taps, int cin, int cin_pad, void *out, hipStream_t s);
int launch_tile_conv(const TileConvLaunch &d, hipStream_t s);
// on-demand correlation (csrc/corr_ondemand.hip): pooled feature pyramid + lookup without a stored volume
int launch_fmap_pyramid(const float *f2, int P, int C, int h, int w, float *const lvl[3], hipStream_t s);
int launch_corr_ondemand(const float *f1, const float *const f2lvl[4], const float *coords, int P, int h, int w,
                         float *out, int ld_out, hipStream_t s);
int launch_convex_upsample(const float *flow_lr, const float *ou, int ld_ou, const float *mask,
                           int P, int h, int w, int pl, int pr, int pt, int pb,
                           float *flow, float *occl, float *sigma, float *packed, hipStream_t s, unsigned *nonfinite = nullptr);

}  // namespace mftx
