// MFMA implicit-GEMM convolution / NT-GEMM for gfx950 (MI355X): fp32 products either on the fp32 matrix instructions or,
// by default, as three fp16 MFMAs on split operands (enum Arith below).  What this header describes is common to both;
// the split-arithmetic K loop (LDS ring of 2-4 chunks, peeled, DMA pieces and operand splits slotted between the MFMAs)
// and the vectorised, branch-free epilogue are described where they stand.
//
//   out[m][n] = epilogue( sum_{tap,c} A[cell(m)+off(tap)][c] * W[n][tap][c] + bias[n] )
//
// A is pixel-major (NHWC) so the reduction axis is contiguous for both
// operands; the same kernel serves
//   * every conv of the RAFT update block and the OU heads (core/update.py),
//   * the all-pairs correlation volume f1 . f2^T / sqrt(C) (core/corr.py:53-69),
//     with kh = kw = 1 and "weights" = the second feature map.
//
// Structure (256 threads = 4 waves, block tile BM x BN, K step 32):
//   * operands come in through raw BUFFER loads: the conv halo, ragged channel
//     counts and the M/N tails are expressed as an out-of-range offset, which
//     the hardware returns as zeros -- no branches, no selects in the K loop;
//     per-row offsets are recomputed once per filter tap, per K step they only
//     advance by 128 bytes;
//   * the loads are LDS-DMA (`buffer_load_dwordx4 ... lds`): no staging VGPRs and
//     no ds_write pass.  An LDS-DMA lands lane-linear (1 KiB per wave
//     instruction = 8 rows of 32 floats), so rows cannot be padded; instead the
//     16-byte chunk c of row r is stored at chunk c ^ ((r >> 1) & 7) -- applied to
//     the SOURCE address on the way in and to the ds_read_b128 address on the way
//     out -- which makes the fragment reads bank-conflict free;
//   * each wave owns (BM/WM) x (BN/WN) as 32x32 v_mfma_f32_32x32x2_f32 tiles
//     (exact fp32, 157 TF peak).  A lane reads 4 consecutive k of its row; lanes
//     0-31 take k 0..3 and lanes 32-63 k 4..7 of an 8-wide group -- a legal
//     permutation of the reduction order as long as A and B use the same one,
//     and the same for every tile shape (results do not depend on the tiling);
//   * software pipeline: fragments are double buffered in registers, the next
//     tile is written to LDS and the barrier taken BEFORE the last 8-wide k
//     group's MFMAs, so LDS latency and barrier skew hide under matrix work.
#include "common.h"
#include "profile.h"
#include <cstdlib>
#include <type_traits>

namespace mftx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// Arithmetic of the matrix products (template parameter AR):
//   AR_F32  : v_mfma_f32_32x32x2_f32, exact fp32 products, 157 TF peak;
//   AR_SPLIT: every fp32 operand x is split into two halves, x = hi + lo / 2048 with hi = fp16(x) and
//             lo = fp16((x - hi) * 2048) (|x - hi - lo / 2048| <= 2^-24 |x|: half an fp32 ulp), and
//             a b = hi_a hi_b + (hi_a lo_b + lo_a hi_b) / 2048 + O(2^-24 |a b|) is formed by three
//             v_mfma_f32_32x32x16_f16 (2.5 PF peak, fp32 accumulation; each fp16 product is exact in fp32).
//             The cross terms have their own accumulator, scaled once in the epilogue: the lo halves stay
//             normal fp16 numbers whatever the magnitude of x.  Operands must be below 65504 in magnitude.
//   AR_PRESPLIT: the same products, with the A operand ALREADY stored in split form by its producer (every 8 channels as
//             [hi x 8 | lo x 8], the layout of the split weights): no VALU work in the K loop at all.  Internal to
//             the refinement engine (mftx_conv_desc.arith = MFTX_ARITH_SPLIT + a_split).
enum Arith { AR_F32 = 0, AR_SPLIT = 1, AR_PRESPLIT = 2 };

// Tuning builds only (-DMFTX_TIMING): per-phase cycle totals of the split K loop, summed over all waves
// (s_memtime stamps; read back with mftx_debug_timing from tools/conv_phase_timing.py)
#ifdef MFTX_TIMING
__device__ unsigned long long mftx_dbg[16];
#endif
#if defined(MFTX_TIMING) && MFTX_TIMING == 1       // (2: tile phases only -- prologue / K loop / epilogue, the loop itself unperturbed)
#define STAMP(i) asm volatile("s_memtime %0" : "=s"(ts[i]))
#define MFTX_TIMING_LOOP 1
#else
#define STAMP(i)
#endif

// hi / lo halves of 8 consecutive k of an activation row, 2.5 instructions per value: v_cvt_pk_f16_f32 for two
// (round to nearest), the exact residual as one mixed-precision fma each (x - hi, hi read as fp16), and the scaled
// low half as v_fma_mixlo/mixhi_f16 (r * 2048 rounded to fp16 into one half of the destination).  Written as one
// assembly block: the compiler's own selection for this arithmetic takes 4 instructions per value, and does not
// know the mixed forms.  The block ends with the two wait states a VALU result needs before an MFMA reads it
// (the hazard recognizer does not see into inline assembly).
__device__ __forceinline__ void split8(const f32x4 &u, const f32x4 &v, float k2048, f16x8 &hi, f16x8 &lo) {
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    float r0, r1, r2, r3, r4, r5, r6, r7;
    asm("v_cvt_pk_f16_f32 %0, %16, %17\n\t"
        "v_cvt_pk_f16_f32 %1, %18, %19\n\t"
        "v_cvt_pk_f16_f32 %2, %20, %21\n\t"
        "v_cvt_pk_f16_f32 %3, %22, %23\n\t"
        "v_fma_mix_f32 %8, %0, -1.0, %16 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %9, %0, -1.0, %17 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %10, %1, -1.0, %18 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %11, %1, -1.0, %19 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %12, %2, -1.0, %20 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %13, %2, -1.0, %21 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %14, %3, -1.0, %22 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %15, %3, -1.0, %23 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %4, %8, %24, 0\n\t"
        "v_fma_mixlo_f16 %5, %10, %24, 0\n\t"
        "v_fma_mixlo_f16 %6, %12, %24, 0\n\t"
        "v_fma_mixlo_f16 %7, %14, %24, 0\n\t"
        "v_fma_mixhi_f16 %4, %9, %24, 0\n\t"
        "v_fma_mixhi_f16 %5, %11, %24, 0\n\t"
        "v_fma_mixhi_f16 %6, %13, %24, 0\n\t"
        "v_fma_mixhi_f16 %7, %15, %24, 0\n\t"
        "s_nop 1"
        : "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3),
          "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
        : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "s"(k2048));
    hi = __builtin_bit_cast(f16x8, u32x4{h0, h1, h2, h3});
    lo = __builtin_bit_cast(f16x8, u32x4{l0, l1, l2, l3});
}

constexpr int BK = 32;
constexpr int LDK = BK;                  // LDS row (floats): unpadded, XOR-swizzled 16-byte chunks
constexpr unsigned OOB = 0x80000000u;    // buffer offset that is out of range for every operand (< 2 GiB)

enum Epi { EPI_GENERIC = 0, EPI_RELU = 1, EPI_GRU_ZR = 2, EPI_GRU_Q = 3,
           EPI_VOLUME = 4 };   // = generic; its own instance so that profiles can tell the correlation volume apart

struct ConvArgs {
    const float *a0, *a1;
    int lda0, lda1, c0, c1;
    const float *w;
    const float *bias;
    const float *addend; int ld_addend;   // optional per-output partial sum (pre-computed inp part of a GRU gate)
    float *out;
    int ldo;
    int M, N, h, wd, kh, kw, cin_pad;
    int stride, hin, win, pad_y, pad_x;   // input grid (hin x win cells per image), conv stride and padding
    int residual_mode;                    // 1: out = relu(act(.) + addend) (residual block tail) instead of a pre-activation addend
    int w_rows;               // valid rows of the W operand
    int act;
    int arith;                // Arith
    int a_pre;                // split arithmetic: both A segments are stored in split form (-> AR_PRESPLIT kernels)
    float out_scale;
    int batch;                                  // correlation volume: one GEMM per pair
    long long a_bstride, w_bstride, o_bstride;  // per batch element
    unsigned a0_bytes, a1_bytes, w_bytes;       // buffer extents (per batch element)
    float *hx; int ld_hx; float *z; float *rh;  // GRU epilogues
    float *hf; int ld_hf;                       // split arithmetic: the fp32 copy of h the gate algebra reads and writes (hx itself may be in split form)
    int out_split;                              // outputs that feed GEMMs (out, rh, hx) are written in split form
    // EPI_VOLUME only: feature grid and the pooled levels (pyramid layout of common.h; out = level 0)
    int vh, vw, sbw, wb0;
    float *lvl1, *lvl2, *lvl3;
    long long s0, s1, s2, s3;                   // floats per query cell, per level
};

// The same for two values, as the piece that is slotted between two MFMAs of the K loop (5 instructions: they issue
// in the shadow of one 32-cycle MFMA).  No trailing wait states: the halves are consumed at least two MFMAs later.
__device__ __forceinline__ void split_pair(float x0, float x1, float k2048, unsigned &h, unsigned &l) {
    float r0, r1;
    asm("v_cvt_pk_f16_f32 %0, %4, %5\n\t"
        "v_fma_mix_f32 %2, %0, -1.0, %4 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %3, %0, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %2, %6, 0\n\t"
        "v_fma_mixhi_f16 %1, %3, %6, 0"
        : "=&v"(h), "=&v"(l), "=&v"(r0), "=&v"(r1)
        : "v"(x0), "v"(x1), "s"(k2048));
}

// Four consecutive channels nb .. nb + 3 (nb % 4 == 0) of a SPLIT-format row: their fp16 high halves are 8 bytes at
// group (nb >> 3), half ((nb >> 2) & 1); the low halves 16 bytes further (common.h: split_row_offset)
__device__ __forceinline__ void store_split4(float *row, int nb, const f32x4 &o, int valid = 4) {
    unsigned h0, h1, l0, l1;
    const float k2048 = 2048.f;
    split_pair(o[0], o[1], k2048, h0, l0);
    split_pair(o[2], o[3], k2048, h1, l1);
    char *dst = reinterpret_cast<char *>(row) + split_row_offset(nb);
    if (valid >= 4) {
        *reinterpret_cast<uint2 *>(dst) = make_uint2(h0, h1);
        *reinterpret_cast<uint2 *>(dst + 16) = make_uint2(l0, l1);
    } else {                                         // N tail: the other channels of the group belong to someone else
        const unsigned short hs[4] = {(unsigned short)h0, (unsigned short)(h0 >> 16), (unsigned short)h1, (unsigned short)(h1 >> 16)};
        const unsigned short ls[4] = {(unsigned short)l0, (unsigned short)(l0 >> 16), (unsigned short)l1, (unsigned short)(l1 >> 16)};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < valid) {
                reinterpret_cast<unsigned short *>(dst)[e] = hs[e];
                reinterpret_cast<unsigned short *>(dst + 16)[e] = ls[e];
            }
    }
}

// Tuning builds only (-DMFTX_ABLATE=n): 1 no global loads, 2 + no barriers, 3 + no LDS reads; 4 no W loads, 5 no A loads, 6 A loads for every fifth chunk only.  A
// compile-time switch on purpose: as run-time branches around the ds_reads these made the compiler
// lose count of the outstanding LDS operations and wait for ALL of them (lgkmcnt(0)) in front of
// every MFMA group.
#ifndef MFTX_ABLATE
#define MFTX_ABLATE 0
#endif
// split-arithmetic loop (timing only, results are garbage): bit 0 operands taken as if they arrived split, 1 no
// refills after the prologue, 2 no waits / barriers, 3 no LDS reads, 4 no MFMAs, 5 no correlation-volume epilogue (the compiler then
// drops the MFMAs too), 6 correlation volume: N tiles innermost
#ifndef MFTX_SABL
#define MFTX_SABL 0
#endif

// Gate non-linearities of the GRU epilogues on the hardware exponential (v_exp_f32, ~1 ulp) and
// reciprocal: 16 values per lane and tile, where libm's expf / tanhf cost ~7 % of the q-gate kernel.
// Absolute error < 2e-7 on outputs in (-1, 1) -- four orders below the parity tolerance; the same
// code runs for every tile shape and batch, so results stay independent of both.
__device__ __forceinline__ float fast_sigmoid(float s) { return __frcp_rn(1.f + __expf(-s)); }
__device__ __forceinline__ float fast_tanh(float s) {
    const float t = __expf(-2.f * fabsf(s));            // in (0, 1]: no overflow
    return copysignf((1.f - t) * __frcp_rn(1.f + t), s);
}

// h <- (1 - z) h + z q, with the contraction spelled out: the scalar and the vectorised epilogues (and every tile shape) must
// round alike, whatever the compiler would fuse in each
__device__ __forceinline__ float gru_blend(float z, float h, float q) { return __fmaf_rn(z, q, __fmul_rn(__fsub_rn(1.f, z), h)); }

__device__ __forceinline__ float act_fn(float v, int act) {
    switch (act) {
        case 1: return relu_keep_nan(v);
        case 2: return 1.f / (1.f + expf(-v));
        case 3: return tanhf(v);
        default: return v;
    }
}

__device__ __forceinline__ f32x4 buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}

// 16 bytes per lane straight into LDS: the wave's 64 lanes land lane-linear at `dst` (wave-uniform);
// an out-of-range offset stores zeros.
// soff: wave-uniform byte offset added to the address (not part of the range check, which is on voff alone)
__device__ __forceinline__ void buf_load_lds(__amdgpu_buffer_rsrc_t r, float *dst, unsigned voff, unsigned soff = 0) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)dst, 16, voff, soff, 0, 0);
}

// s_barrier with compiler fences on both sides (the intrinsic alone does not order LDS accesses)
__device__ __forceinline__ void block_barrier() {
    // s_waitcnt lgkmcnt(0): gfx950 has back-off barriers, so the compiler inserts NO wait in front of s_barrier and the builtin is no
    // fence -- without this a wave's last ds_write may still sit in the LDS queue when another wave reads the slot behind the
    // barrier (found in round 5 with tools/race_kernels.py: harmless with the GPU to itself, wrong values under contention).
    // LDS only: global prefetches and LDS-DMA loads (vmcnt) stay in flight, their consumers count them themselves.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {   // counted wait: leaves N LDS-DMA loads in flight
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Waves per SIMD the register allocator must leave room for (second __launch_bounds__ argument):
// left alone it spends registers freely in the pinned K loop and costs a resident workgroup.
#ifndef MFTX_MINW1
#define MFTX_MINW1 4      // 128 registers: nothing spills; 5 (96 registers, spills in the GRU epilogues) measures 0.5 % slower
#endif
constexpr int min_waves(int wave_tiles, int epi = 0) {
    // (the q-gate epilogue keeps z and h of a tile in registers: at 128 it spills 34 of them)
    return wave_tiles == 1 ? (epi == 3 ? 3 : MFTX_MINW1) : wave_tiles == 2 ? 3 : 2;
}
// split arithmetic: two accumulator sets and raw + split fragments
constexpr int min_waves_split(int wave_tiles, int waves, int mt = 32) { return mt == 16 ? 1 : waves == 8 ? 2 : wave_tiles == 1 ? 3 : 2; }

// Epilogue of the correlation-volume GEMM: one wave holds 32 query rows x one super-block (2 x 2 blocks of
// 8 x 4 target cells, 128 columns) in acc[4].  The tile goes through the wave's private 16 KiB of LDS once:
//   level 0: read back as float4 -> 512-byte runs per query (global_store_dwordx4, 16 instead of 64 stores);
//   level 1: 2 x 2 means of level 0 in ATen's order ((a + b) + c + d) * 0.25 -> exactly one 8 x 4 block per
//            query, stored as a full 128-byte line, and kept in LDS for
//   level 2 (4 x 2 per super-block) and level 3 (2 x 1), formed the same way from the level below --
// bit-identical to avg_pool2d applied level by level (core/corr.py:26-28); no second pass over the volume.
__device__ __forceinline__ float pool4(const float2 top, const float2 bot) {
    return (((top.x + top.y) + bot.x) + bot.y) * 0.25f;
}

__device__ __forceinline__ void volume_epilogue(const ConvArgs &p, const f32x16 (&acc)[4], float *st, int lane,
                                                int q0, int sb, int bz) {
    const int Nq = p.M;                                       // query cells per pair
    const int sby = sb / p.sbw, sbx = sb - sby * p.sbw;
    // ---- stage: st[row][128], row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column = 32 j + (lane & 31)
    {
        float *w = st + (4 * (lane >> 5)) * 128 + (lane & 31);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                w[((r & 3) + 8 * (r >> 2)) * 128 + 32 * j] = acc[j][r] * p.out_scale;
    }
    const long long qbase = (long long)bz * Nq;
    // ---- level 0: float4 pieces; piece c4 of a row = block (c4 >> 3), floats 4 (c4 & 7) .. +3 of it
    {
        float *out0 = p.out;
#pragma unroll
        for (int t0 = 0; t0 < 16; t0 += 4) {          // four LDS reads in flight, then their four stores
            f32x4 v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = *reinterpret_cast<const f32x4 *>(st + ((t0 + t) * 64 + lane) * 4);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int f = (t0 + t) * 64 + lane, row = f >> 5, c4 = f & 31, jb = c4 >> 3;
                const int q = q0 + row;
                if (q < Nq) {
                    float *dst = out0 + (qbase + q) * p.s0 +
                                 ((long long)(2 * sby + (jb >> 1)) * p.wb0 + 2 * sbx + (jb & 1)) * 32 + (c4 & 7) * 4;
                    *reinterpret_cast<f32x4 *>(dst) = v[t];
                }
            }
        }
    }
    // ---- level 1: 16 values per lane; lanes of a half-wave = the 32 cells of the level-1 block of one query
    float l1[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int e = t * 64 + lane, row = e >> 5, pos = e & 31, y1 = pos >> 3, x1 = pos & 7;
        const float *s = st + row * 128 + ((y1 >> 1) * 2 + (x1 >> 2)) * 32 + (y1 & 1) * 16 + (x1 & 3) * 2;
        l1[t] = pool4(*reinterpret_cast<const float2 *>(s), *reinterpret_cast<const float2 *>(s + 8));
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int e = t * 64 + lane, row = e >> 5, pos = e & 31;
        const int q = q0 + row;
        if (q < Nq) p.lvl1[(qbase + q) * p.s1 + ((long long)sby * p.sbw + sbx) * 32 + pos] = l1[t];
    }
    // level 1 -> LDS [row][32] (overlays the level-0 stage: this wave's reads of it are complete, LDS
    // operations of one wave execute in order)
#pragma unroll
    for (int t = 0; t < 16; ++t) st[t * 64 + lane] = l1[t];
    // ---- level 2: 4 x 2 per query, 4 values per lane; kept at st[1024 + row * 8 + ..]
    const int h2 = p.vh >> 2, w2 = p.vw >> 2, h3 = p.vh >> 3, w3 = p.vw >> 3;
    float l2[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int e = t * 64 + lane, row = e >> 3, y2 = (e >> 2) & 1, x2 = e & 3;
        const float *s = st + row * 32 + y2 * 16 + x2 * 2;
        l2[t] = pool4(*reinterpret_cast<const float2 *>(s), *reinterpret_cast<const float2 *>(s + 8));
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int e = t * 64 + lane, row = e >> 3, y2 = 2 * sby + ((e >> 2) & 1), x2 = 4 * sbx + (e & 3);
        const int q = q0 + row;
        if (q < Nq && y2 < h2 && x2 < w2) p.lvl2[(qbase + q) * p.s2 + (long long)y2 * w2 + x2] = l2[t];
        st[1024 + e] = l2[t];
    }
    // ---- level 3: 2 x 1 per query, one value per lane
    {
        const int row = lane >> 1, x3l = lane & 1;
        const float *s = st + 1024 + row * 8 + x3l * 2;
        const float v = pool4(*reinterpret_cast<const float2 *>(s), *reinterpret_cast<const float2 *>(s + 4));
        const int q = q0 + row, x3 = 2 * sbx + x3l;
        if (q < Nq && sby < h3 && x3 < w3) p.lvl3[(qbase + q) * p.s3 + (long long)sby * w3 + x3] = v;
    }
}

// MT = 32: each wave owns 32x32 outputs per MFMA tile (v_mfma_f32_32x32x2_f32).  MT = 16: 16x16 outputs
// (v_mfma_f32_16x16x4_f32), for small M: a 32x64 workgroup tile still keeps 8 waves -- every SIMD -- busy,
// so M x N splits into twice or four times as many workgroups when 64x64 tiles would leave CUs idle (one
// flow pair per GPU: 64 row tiles; the N = 128 layers reach 128 of 256 CUs).  Fed the k sequence the 32x32x2
// form consumes -- (k, k+4, k+1, k+5), (k+2, k+6, k+3, k+7) across its four lane groups -- the 16x16x4
// form rounds bit-identically (both are sequential fmaf chains; tools/micro/mfma_order.hip), so the
// tile choice still does not show in the results.
// WS = 1 (split arithmetic, A pre-split): WARP-SPECIALISED workgroup -- WM x WN consumer waves (fragment reads and MFMAs,
// nothing else in their K loop) and as many producer waves (all of the LDS-DMA staging: an LDS-DMA piece holds the wave
// that issues it for ~70 cycles, a third of a chunk's matrix work when the same wave also feeds the matrix pipe).  Same
// ring, same barrier per chunk, same fragment order: same bits.
template <int BM, int BN, int WM, int WN, int EPI, int MT = 32, int AR = AR_F32, int NS = 2, int WS = 0>
__device__ __forceinline__ void conv_gemm_body(const ConvArgs &p, int vt0) {
    static_assert(!WS || (AR == AR_PRESPLIT && MT == 32), "warp-specialised form: split arithmetic with a pre-split A, 32-row MFMA tiles");
    static_assert(AR == AR_F32 || MT == 32 || AR == AR_PRESPLIT, "split arithmetic: 32x32 MFMA tiles; 16x16 for a pre-split A only");
    // NS: LDS ring of K chunks.  fp32 MFMA: two (a chunk is > 1000 matrix cycles per wave, deeper rings were
    // measured: no gain).  Split arithmetic: a chunk is 192 matrix cycles per MFMA tile, well below the L2 latency:
    // three or four chunks are kept in flight.
    static_assert(AR != AR_F32 || NS == 2, "ring depth");
    constexpr bool SPLIT = AR != AR_F32;          // split arithmetic, A split in registers (AR_SPLIT) or by its producer (AR_PRESPLIT)
    constexpr bool PRE = AR == AR_PRESPLIT;
    // BM need not be a multiple of WM MT: the last row of waves then owns fewer MFMA row tiles (224 x 128: four waves of
    // 4 x 1 tiles over four of 3 x 1 -- paired per SIMD, 7 tiles each; 7 x 4096 cells = 128 x 224, two column tiles: 256 tiles)
    constexpr int BMW = (BM + WM * MT - 1) / (WM * MT) * (WM * MT);
    static_assert((BMW - BM) % MT == 0, "tile rows: whole MFMA tiles");
    constexpr int TM = BMW / WM / MT, TN = BN / WN / MT;
    constexpr int NR = MT == 32 ? 16 : 4;       // accumulator registers per MFMA tile
    constexpr int RPP = 8 * WM * WN;            // rows staged per pass: 8 per wave (one 1 KiB LDS-DMA)
    constexpr int RA = (BM + RPP - 1) / RPP, RB = (BN + RPP - 1) / RPP;   // (a small tile leaves some waves without A rows)
    static_assert(MT == 32 || MT == 16, "MFMA tile");
    static_assert(TM >= 1 && TN >= 1 && RA >= 1 && RB >= 1, "tile");
    static_assert(RPP % 16 == 0, "the chunk swizzle must not depend on the pass");
    constexpr int BMS = RA * RPP, BNS = RB * RPP;   // rows per ring slot: whole staging passes (112-row tile: 128, the last 16 zero-filled and never read)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                       // [NS][BMS][LDK]
    float *Bs = smem + NS * BMS * LDK;      // [NS][BNS][LDK]

    const int lane = threadIdx.x & 63;
    const int wid_all = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // scalar: LDS-DMA destinations live in m0
    const bool is_prod = WS && wid_all >= WM * WN;             // (wave-uniform) a producer wave of the warp-specialised form
    const int wid = is_prod ? wid_all - WM * WN : wid_all;     // consumer: which wave tile; producer: which staging rows
    const int tid = wid * 64 + lane;
    const int wm = wid / WN, wn = wid % WN;
    const int srow = tid >> 3;  // 0..RPP-1: staging row of this lane within an RPP-row group
    // this lane's LDS chunk (tid & 7) receives logical chunk (tid & 7) ^ swz(row); rows advance by
    // RPP (a multiple of 16) per group, so swz = (row >> 1) & 7 depends on srow only
    // (16-row MFMA tiles read with another lane -> row mapping and carry their own swizzle, below)
    auto swz = [](int row) {
        const int s2 = (row >> 1) & 7;
        return (SPLIT && MT == 16) ? s2 ^ ((((s2 >> 1) ^ (s2 >> 2)) & 1) << 1) : s2;
    };
    const int col4 = ((tid & 7) ^ swz(srow)) * 4;

    // Persistent workgroups with an XCD-aware tile order.  The hardware dispatcher
    // fills free slots greedily, which leaves whole CUs idle in a ragged last
    // round (7 * 2^k tiles on 2^k slots); a fixed round-robin keeps every CU
    // equally loaded.  Workgroup b runs on XCD b % 8 (observed; only speed depends
    // on it), so virtual tile v is decoded as xcd = v % 8, and each XCD walks its
    // own contiguous band of M tiles with the N tiles innermost: the column tiles
    // of one A tile and the 3x3 halo rows of neighbouring A tiles then meet in
    // the same 4 MiB L2 instead of being fetched once per XCD.
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int band = (tiles_m + 7) / 8;               // M tiles per XCD
    const int per_batch = band * tiles_n;              // virtual tiles per XCD per batch element
    const int n_virtual = 8 * per_batch * p.batch;
    const int taps = p.kh * p.kw;
    const int cpt = p.cin_pad / BK;          // K chunks per tap
    const int T = taps * cpt;
    const unsigned ktot_b = (unsigned)taps * p.cin_pad * 4u;
    const int ctot = p.c0 + p.c1;
    const int hw = p.h * p.wd;
    // A tap's chunks form up to three RUNS inside which a chunk differs from the previous one only
    // by +128 B in every offset: segment 0, segment 1 (second input tensor of a concatenation), and
    // -- when the channel count is not a multiple of 32 -- the last, partly zero-filled chunk.
    const int seg_cc = p.c1 > 0 ? p.c0 / BK : cpt;           // first chunk of segment 1
    const int rag_cc = (ctot % BK) ? cpt - 1 : cpt;          // the ragged chunk, if any

    const int tm_act = wm == WM - 1 ? TM - (BMW - BM) / MT : TM;      // this wave's live row tiles (wave-uniform)
    const int a_row0 = wm * TM * MT + (lane & (MT - 1));
    const int b_row0 = wn * TN * MT + (lane & (MT - 1));
    // 16-byte chunk of an 8-wide k group this lane reads: MT = 32: lanes 0-31 chunk 2kk, lanes 32-63 chunk
    // 2kk+1, all four values used; MT = 16: lane group g = lane >> 4 reads chunk 2kk + (g & 1) and uses
    // elements (g >> 1) and (g >> 1) + 2 of it
    const int khalf = MT == 32 ? lane >> 5 : (lane >> 4) & 1;
    const bool hi_pair = MT == 16 && (lane >> 5) != 0;
    const int a_sw = swz(a_row0), b_sw = swz(b_row0);   // same for every 16- / 32-row step
    // fragment addresses of the four 8-wide k groups (the XOR swizzle is not additive: one each)
    const float *a_frag[4], *b_frag[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        a_frag[kk] = As + a_row0 * LDK + (((kk * 2 + khalf) ^ a_sw) * 4);
        b_frag[kk] = Bs + b_row0 * LDK + (((kk * 2 + khalf) ^ b_sw) * 4);
    }
    // split arithmetic: a lane reads the 8 consecutive k (two 16-byte chunks) of its row and of its half of a 16-wide
    // k group g = 0, 1 of the chunk: chunks 4 g + 2 (lane >> 5) + {0, 1}
    const float *a_frag2[2][2], *b_frag2[2][2];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            a_frag2[g][c] = As + a_row0 * LDK + (((4 * g + 2 * khalf + c) ^ a_sw) * 4);
            b_frag2[g][c] = Bs + b_row0 * LDK + (((4 * g + 2 * khalf + c) ^ b_sw) * 4);
        }
    // Split arithmetic on 16x16x32 MFMAs (A pre-split).  A row of a chunk is 8 pieces of 16 bytes: h0 l0 h1 l1 h2 l2 h3 l3,
    // the high / low halves of channels 8 g .. 8 g + 7.  Lane quarter q = lane >> 4 supplies k slots 8 q .. 8 q + 7 of an
    // instruction, and the three products' operands are plain ds_read_b128 with a per-quarter piece:
    //   operand 0 (hi hi, k = all 32 channels):          h_q                              piece 2 q
    //   operand 1 (cross terms of channels  0..15):  A: h0 h1 l0 l1   W: l0 l1 h0 h1      (slots 0..15: hi_a lo_b, 16..31: lo_a hi_b)
    //   operand 2 (cross terms of channels 16..31):  A: h2 h3 l2 l3   W: l2 l3 h2 h3
    // Swizzle: ds_read_b128 is served in groups of 16 lanes = rows {0-3, 12-15} of one quarter and {4-11} of the next
    // (or the other way round), whose pieces differ by 2; s ^ 2 for row pairs s = 2..5 spreads each group over all 64 banks.
    const float *a_frag16[3], *b_frag16[3];
    {
        const int q = lane >> 4;
        const int pa[3] = {2 * q, 2 * (q & 1) + (q >> 1), 4 + 2 * (q & 1) + (q >> 1)};
        const int pb[3] = {2 * q, 2 * (q & 1) + 1 - (q >> 1), 4 + 2 * (q & 1) + 1 - (q >> 1)};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a_frag16[c] = As + a_row0 * LDK + ((pa[c] ^ a_sw) * 4);
            b_frag16[c] = Bs + b_row0 * LDK + ((pb[c] ^ b_sw) * 4);
        }
    }
    // LDS-DMA destinations: wave `wid` fills rows [RPP i + 8 wid, +8) of each RPP-row group (1 KiB, lane-linear)
    float *const a_dst = As + wid * 8 * LDK;
    float *const b_dst = Bs + wid * 8 * LDK;

    for (int vt = vt0; vt < n_virtual; vt += gridDim.x) {
    const int xcd = vt & 7;
    const int q = vt >> 3;
    const int bz = q / per_batch;
    const int r = q - bz * per_batch;
    // conv layers: N tiles innermost (they share the gathered A tile); correlation volume: M tiles innermost --
    // the B tile (128 target rows of f2) stays put while the XCD's band of A tiles (a few hundred KB of f1)
    // cycles through its L2, instead of every XCD re-reading all of f2 once per M tile (fabric reads 879 -> ~270 MB)
    constexpr bool VOL_M_INNER = EPI == EPI_VOLUME && !(MFTX_SABL & 64);
    const int tm = VOL_M_INNER ? xcd * band + r % band : xcd * band + r / tiles_n;
    if (tm >= tiles_m) continue;                       // ragged band (uniform per workgroup)
    const int m0 = tm * BM;
    const int n0 = (VOL_M_INNER ? r / band : r % tiles_n) * BN;
    float *out = p.out + bz * p.o_bstride;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    block_barrier();                      // previous tile's last LDS reads are done
#ifdef MFTX_TIMING
    unsigned long long tt0, tt1 = 0, tt2 = 0;       // tile phases: [8] prologue, [9] K loop, [10] epilogue, [11] tiles
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt0));
#endif

    const __amdgpu_buffer_rsrc_t rA0 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.a0 + bz * p.a_bstride), 0, p.a0_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rA1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.a1 ? p.a1 : p.a0), 0, p.a1_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.w + bz * p.w_bstride), 0, p.w_bytes, 0x00020000);

    // ---- per-thread staging coordinates
    int ay[RA], ax[RA], am[RA];              // input-grid origin of the row's window, and its image's first cell
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        const int m = m0 + srow + RPP * i;
        if (m < p.M && srow + RPP * i < BM) {
            const int img = m / hw;
            const int rem = m - img * hw;
            const int oy = rem / p.wd;
            ay[i] = oy * p.stride - p.pad_y;
            ax[i] = (rem - oy * p.wd) * p.stride - p.pad_x;
            am[i] = img * p.hin * p.win;
        } else {
            ay[i] = -100000; ax[i] = -100000; am[i] = 0;
        }
    }
    unsigned woff[RB];                       // byte offsets into W, +128 B per chunk (W is [n][tap][cin_pad])
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        if constexpr (EPI == EPI_VOLUME) {
            // N tile = one super-block (16 x 8 target cells) of the blocked level 0: tile column c holds
            // block (c >> 6, (c >> 5) & 1) of the 2 x 2, cell (y, x) = ((c & 31) >> 3, c & 7) inside it --
            // each 32-column MFMA tile is one 8 x 4 block, i.e. one 128-byte line of the output per query.
            const int sb = n0 / BN, c = srow + RPP * i;
            const int ty = 8 * (sb / p.sbw) + 4 * (c >> 6) + ((c & 31) >> 3);
            const int tx = 16 * (sb % p.sbw) + 8 * ((c >> 5) & 1) + (c & 7);
            woff[i] = (ty < p.vh && tx < p.vw) ? (unsigned)(ty * p.vw + tx) * ktot_b + col4 * 4u : OOB;   // outside: zeros
        } else {
            const int n = n0 + srow + RPP * i;
            woff[i] = (n < p.w_rows && srow + RPP * i < BN) ? (unsigned)n * ktot_b + col4 * 4u : OOB;
        }
    }
    // Byte offsets of the NEXT chunk to fetch.
    //  * split arithmetic: per row and channel segment (a0 / a1 of a concatenated input), computed once per filter tap
    //    for both segments, +128 B per chunk after that -- a segment change inside a tap costs nothing (recomputing
    //    every row's offset there, ~100 instructions in front of the chunk's DMA, showed in full with a ring of two
    //    chunks and 0.4 us of matrix work per chunk);
    //  * fp32 MFMA (1 us of matrix work per chunk hides it; fewer instructions and registers per chunk count): per
    //    RUN -- a tap's chunks form up to three runs inside which every offset advances by 128 B: segment 0, segment 1,
    //    and the last, partly zero-filled chunk when the channel count is not a multiple of 32.
    unsigned acur0[RA], acur1[SPLIT ? RA : 1];
    unsigned acell[SPLIT ? 1 : RA];          // fp32: this tap's source cell of each row, or OOB (conv halo / M tail)
    __amdgpu_buffer_rsrc_t rA = rA0;
    int cc = 0, dy = 0, dx = 0, left = 0;    // next chunk to fetch: channel chunk, filter tap (dy, dx); fp32: chunks left in the run
    // ragged last chunk (channel count not a multiple of 32): this lane's piece lies beyond the last channel -> zeros
    // (pre-split A: a 16-byte piece holds one half of 8 channels)
    const bool rag_dead = (cpt - 1) * BK + (PRE ? (col4 & ~7) : col4) >= ctot;
    auto new_tap = [&]() {                   // split arithmetic: acur0 = the offsets in use, acur1 = where segment 1 will start
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int yy = ay[i] + dy, xx = ax[i] + dx;
            const bool ok = (yy >= 0) & (yy < p.hin) & (xx >= 0) & (xx < p.win);
            const unsigned cell = (unsigned)(am[i] + yy * p.win + xx);
            acur0[i] = ok ? (cell * (unsigned)p.lda0 + col4) * 4u : OOB;      // conv halo / M tail: out of range -> zeros
            if constexpr (SPLIT) acur1[i] = ok ? (cell * (unsigned)p.lda1 + col4) * 4u : OOB;
        }
    };
    auto next_run = [&]() {                  // fp32 MFMA: (tap, cc) starts a run, offsets from scratch
        if constexpr (!SPLIT) {
            if (cc == 0) {
#pragma unroll
                for (int i = 0; i < RA; ++i) {
                    const int yy = ay[i] + dy, xx = ax[i] + dx;
                    const bool ok = (yy >= 0) & (yy < p.hin) & (xx >= 0) & (xx < p.win);
                    acell[i] = ok ? (unsigned)(am[i] + yy * p.win + xx) : OOB;
                }
            }
            const bool seg1 = cc >= seg_cc;      // wave-uniform
            const bool ragged = cc >= rag_cc;
            const unsigned lda = (unsigned)(seg1 ? p.lda1 : p.lda0);
            const unsigned cb = (unsigned)(cc * BK - (seg1 ? p.c0 : 0)) * 4u;
            const bool lane_ok = !ragged || !rag_dead;
            rA = seg1 ? rA1 : rA0;
#pragma unroll
            for (int i = 0; i < RA; ++i)
                acur0[i] = (acell[i] != OOB && lane_ok) ? (acell[i] * lda + col4) * 4u + cb : OOB;
            const int stop = ragged ? cpt : (seg1 ? rag_cc : (seg_cc < rag_cc ? seg_cc : rag_cc));
            left = stop - cc;
        }
    };
    // global -> LDS (DMA) for the next chunk, in RA + RB pieces of 1 KiB per wave: begin (a new tap's / run's offsets),
    // the pieces -- issued one by one between the MFMAs of the split-arithmetic loop: eight of them back to back
    // fill the address unit's queue and hold the wave (and its MFMAs) for most of a microsecond -- and end (advance)
    auto fetch_begin = [&]() {
        if constexpr (SPLIT) {
            // the per-chunk work is three wave-uniform tests; everything per row happens once per tap, at its first chunk, at
            // the segment change and at the ragged last chunk (the pieces themselves: load, + 128 B)
            if (cc == 0) { new_tap(); rA = rA0; }
            if (cc == seg_cc) {                  // (seg_cc == cpt: one segment, never)
#pragma unroll
                for (int i = 0; i < RA; ++i) acur0[i] = acur1[i];
                rA = rA1;
            }
            if (cc == rag_cc) {                  // a lane whose piece lies beyond the last channel: zeros (the next tap starts afresh)
#pragma unroll
                for (int i = 0; i < RA; ++i) acur0[i] = rag_dead ? OOB : acur0[i];
            }
        } else {
            if (left == 0) next_run();
        }
    };
    auto fetch_piece = [&](int buf, int k) {
        if (k < RA) {
            const unsigned off = acur0[k];
            acur0[k] += BK * 4u;                 // (an OOB offset stays out of range)
            if (MFTX_ABLATE != 5 && (MFTX_ABLATE != 6 || cc % 5 == 0) && (BM % RPP == 0 || SPLIT || wid * 8 + RPP * k < BM))   // (split: counted waits, every wave issues every piece)
                buf_load_lds(rA, a_dst + buf * BMS * LDK + RPP * k * LDK, off);
        } else {
            const int i = k - RA;
            if (MFTX_ABLATE != 4 && (BN % RPP == 0 || SPLIT || wid * 8 + RPP * i < BN))
                buf_load_lds(rW, b_dst + buf * BNS * LDK + RPP * i * LDK, woff[i]);
            woff[i] += BK * 4u;
        }
    };
    auto fetch_end = [&]() {
        if constexpr (!SPLIT) --left;
        if (++cc == cpt) {
            cc = 0;
            if (++dx == p.kw) { dx = 0; ++dy; }
        }
    };
    auto fetch = [&](int buf) {
        fetch_begin();
#pragma unroll
        for (int k = 0; k < RA + RB; ++k) fetch_piece(buf, k);
        fetch_end();
    };

    typedef float acc_t __attribute__((ext_vector_type(NR)));
    acc_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[i][j][r] = 0.f;

    acc_t accx[SPLIT ? TM : 1][SPLIT ? TN : 1];      // cross terms (x 2048)
    if constexpr (SPLIT) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < NR; ++r) accx[i][j][r] = 0.f;
    }
    // ---- split arithmetic: raw fp32 fragments double buffered in registers, split right before their MFMAs
    f32x4 ra[2][SPLIT ? TM : 1][2], rb[2][SPLIT ? TN : 1][2];
    auto read_raw = [&](int buf, int g, int slot) {
        if constexpr (SPLIT && !(MFTX_SABL & 8)) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    ra[slot][i][c] = *reinterpret_cast<const f32x4 *>(a_frag2[g][c] + buf * BMS * LDK + MT * i * LDK);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    rb[slot][j][c] = *reinterpret_cast<const f32x4 *>(b_frag2[g][c] + buf * BNS * LDK + MT * j * LDK);
        }
    };
    const float k2048 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(0x45000000));   // scalar register operand
    // One 16-wide k group out of raw register set `set`: for every A tile of the wave 3 TN MFMAs, and -- slotted
    // between them, pair by pair, in the shadow of the matrix pipe -- the split of the NEXT A tile: tile i + 1 of
    // this group, or tile 0 of the next group (raw set `nset`, when `have_next`).  ah / al hold the split operands
    // of the tile about to be multiplied (tile 0 on entry); the order is pinned with scheduling barriers.
    f16x8 ah[SPLIT ? TM : 1], al[SPLIT ? TM : 1];
    // (have_next, do_refill: compile-time -- as run-time flags they were a branch around every DMA piece and eight selects
    // per chunk; the K loop below is peeled into its steady part, the chunks without refill and the last one instead)
    auto group = [&](int set, int nset, auto have_next_, auto do_refill_, int refill) {
        constexpr bool have_next = decltype(have_next_)::value, do_refill = decltype(do_refill_)::value;
        if constexpr (SPLIT && MT == 32) {
            f16x8 bh[TN], bl[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {           // the weights arrive split: [hi x 8 | lo x 8] per 8 k = the two chunks read
                bh[j] = __builtin_bit_cast(f16x8, rb[set][j][0]);
                bl[j] = __builtin_bit_cast(f16x8, rb[set][j][1]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const bool in_group = i + 1 < TM;
                const bool do_split = (in_group || have_next) && !(MFTX_SABL & 1) && !PRE;
                const f32x4 &u = in_group ? ra[set][i + 1 < TM ? i + 1 : 0][0] : ra[nset][0][0];
                const f32x4 &v = in_group ? ra[set][i + 1 < TM ? i + 1 : 0][1] : ra[nset][0][1];
                u32x4 nh = {0, 0, 0, 0}, nl = {0, 0, 0, 0};
                auto piece = [&](int q) {            // pair q of the split
                    if (do_split) {
                        unsigned hh, ll;
                        split_pair(q < 2 ? u[2 * q] : v[2 * q - 4], q < 2 ? u[2 * q + 1] : v[2 * q - 3], k2048, hh, ll);
                        nh[q] = hh;
                        nl[q] = ll;
                    }
                };
                __builtin_amdgcn_sched_barrier(0);
                // product by product (consecutive MFMAs never wait for each other's accumulator), a pair of the
                // split behind each of the first four
#pragma unroll
                for (int m = 0; m < 3 * TN; ++m) {
                    const int j = m % TN, prod = m / TN;
                    if ((MFTX_SABL & 16) || (BMW != BM && i >= tm_act)) {}
                    else if (prod == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                    else if (prod == 1) accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accx[i][j], 0, 0, 0);
                    else accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], accx[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (m < 4) { piece(m); __builtin_amdgcn_sched_barrier(0); }
                    if (do_refill && i * 3 * TN + m < RA + RB) { fetch_piece(refill, i * 3 * TN + m); __builtin_amdgcn_sched_barrier(0); }   // one DMA piece per MFMA
                }
#pragma unroll
                for (int q = 3 * TN; q < 4; ++q) piece(q);       // (TN = 1: fewer MFMAs than pairs)
                if (i == TM - 1 && do_refill) {
#pragma unroll
                    for (int k = 3 * TN * TM; k < RA + RB; ++k) fetch_piece(refill, k);     // (fewer MFMAs than pieces)
                }
                if ((MFTX_SABL & 1) || PRE) {            // the two chunks read ARE the halves
                    nh = __builtin_bit_cast(u32x4, u);
                    nl = __builtin_bit_cast(u32x4, v);
                }
                if (in_group || have_next) {
                    ah[in_group ? i + 1 : 0] = __builtin_bit_cast(f16x8, nh);
                    al[in_group ? i + 1 : 0] = __builtin_bit_cast(f16x8, nl);
                }
            }
        }
    };
    f32x4 fa[2][TM], fb[2][TN];              // register double buffer of MFMA fragments
    auto read_frags = [&](int buf, int kk, int slot) {
        if (MFTX_ABLATE == 3) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
            fa[slot][i] = *reinterpret_cast<const f32x4 *>(a_frag[kk] + buf * BMS * LDK + MT * i * LDK);
#pragma unroll
        for (int j = 0; j < TN; ++j)
            fb[slot][j] = *reinterpret_cast<const f32x4 *>(b_frag[kk] + buf * BNS * LDK + MT * j * LDK);
    };
    auto mma = [&](int slot) {
        if constexpr (MT == 32) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[slot][i][s], fb[slot][j][s], acc[i][j], 0, 0, 0);
        } else {
            // two instructions per 8-wide k group: lane groups (0,1,2,3) supply k = (0,4,1,5), then (2,6,3,7)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float a = hi_pair ? fa[slot][i][2 * s + 1] : fa[slot][i][2 * s];
                        const float b = hi_pair ? fb[slot][j][2 * s + 1] : fb[slot][j][2 * s];
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i][j], 0, 0, 0);
                    }
        }
    };
    // One K chunk out of ring slot `buf` (compile-time after unrolling: every LDS address is a register
    // plus an immediate).  The other slot was released by the barrier of the previous step, so its
    // refill is issued first and has this whole step of MFMA work to land; fragments are double
    // buffered in registers; the barrier is taken BEFORE the last k group's MFMAs and the next
    // chunk's first fragments are read right behind it, hidden under those MFMAs.
    // The order below is pinned with scheduling barriers: left alone, the compiler sinks every
    // ds_read to just in front of the MFMAs that consume it (one register set instead of two) and
    // the LDS latency shows four times per chunk -- invisible with five waves per SIMD, a third
    // of the time with one (M = 4096, one flow pair per GPU).
    auto step = [&](int buf, bool more, bool more2) {
        read_frags(buf, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(0);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(buf, 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(1);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(buf, 3, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(0);
        __builtin_amdgcn_sched_barrier(0);
        if (more && MFTX_ABLATE != 2 && MFTX_ABLATE != 3) {
            // own LDS reads of `buf` done (lgkmcnt), own DMA of the next chunk landed (vmcnt), then
            // every wave agrees (barrier)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wait_vmcnt<0>();
            block_barrier();
        }
        if (more) read_frags(buf ^ 1, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(1);
        __builtin_amdgcn_sched_barrier(0);
        // slot `buf` is free since the barrier: refill it with the chunk after next -- the DMA issue
        // (the slowest instructions of the loop) runs under the four MFMAs just queued, and the data
        // has most of a step to land
        if (more2 && (MFTX_ABLATE < 1 || MFTX_ABLATE > 3)) fetch(buf);
        __builtin_amdgcn_sched_barrier(0);
    };

    if constexpr (SPLIT && MT == 16) {
        // 16-row MFMA tiles (v_mfma_f32_16x16x32_f16), so that the workgroup tile can be 112 rows: 7 x 4096 cells are
        // 256 x 112 -- one tile for every CU, where 128-row tiles leave 32 of the 256 CUs idle.  ONE wave per SIMD, each
        // owning all TM row tiles x 64 columns (224 accumulator registers): per 32-wide chunk a wave reads 2 TM + 2 TN
        // fragments from LDS for 3 TM TN MFMAs -- half the LDS traffic per matrix cycle of the 64 x 64 waves of the
        // 32-row form.
        //   The instruction rounds exactly like two 32x32x16 ones over k 0..15 and k 16..31 (tools/micro/
        // mfma_f16_order.hip), so results stay independent of the tile shape if every accumulator sees the same
        // sequence: acc += hi hi (k 0..31) as it stands; the cross terms of the 32-row form arrive as hi lo (g 0),
        // lo hi (g 0), hi lo (g 1), lo hi (g 1) for the chunk's two 16-wide groups g -- here one instruction per group
        // whose k slots 0..15 carry (hi_a, lo_b) and 16..31 (lo_a, hi_b) of that group (operands 1 and 2 above).
        static_assert(PRE, "16-row tiles: A arrives split");
        static_assert(NS == 3 && TN == 4 && TM >= 7, "16-row tiles: schedule below");
        // Ring of three chunks, ONE barrier per chunk, taken before row tile BAR of chunk c: it certifies that chunk c + 1
        // has landed (issued a whole chunk earlier) and that every wave is past chunk c - 1, whose slot takes chunk c + 2
        // -- its DMA pieces go out three per row tile behind MFMAs, as do the reads of the next chunk's weight fragments
        // (one column tile per row tile; all four waves reading 12 KiB at once right behind a barrier would hold the LDS
        // port for 200 cycles); the next chunk's first A tile is read under the last row tile's MFMAs.  No VALU work at
        // all in the loop.  With one wave per SIMD nothing else hides latency: every ds_read is issued >= 12 MFMAs ahead.
        constexpr int L = RA + RB, BAR = 2, PPT = (L + 3) / 4;
        fetch(0);
        if (T > 1) fetch(1);
        wait_vmcnt<0>();
        block_barrier();
        f16x8 fb16[2][TN][3];               // weights of this / the next chunk
        f16x8 fa16[2][3];                   // A row tile i in set (i + parity) & 1
        auto read_b = [&](int buf, int j, int set) {
#pragma unroll
            for (int o = 0; o < 3; ++o) fb16[set][j][o] = *reinterpret_cast<const f16x8 *>(b_frag16[o] + buf * BNS * LDK + 16 * j * LDK);
        };
        auto read_a = [&](int buf, int i, int set) {
#pragma unroll
            for (int o = 0; o < 3; ++o) fa16[set][o] = *reinterpret_cast<const f16x8 *>(a_frag16[o] + buf * BMS * LDK + 16 * i * LDK);
        };
#pragma unroll
        for (int j = 0; j < TN; ++j) read_b(0, j, 0);
        read_a(0, 0, 0);
        int slot = 0;
        // Every chunk runs the same straight-line schedule, the last two included: what they read ahead (the ring's next
        // slot) is never used, what they fetch ahead (chunks T, T + 1: offsets past the last filter tap -- valid or
        // out-of-range addresses, 3 % more L2 reads) lands in slots nobody reads, and is waited for before the ring
        // becomes epilogue staging.  (Peeled tails with their own schedules cost the register allocator its mind.)
        auto chunk = [&](auto pa_, auto pb_) {
            constexpr int PA = decltype(pa_)::value, PB = decltype(pb_)::value;
            const int nslot = slot + 1 == NS ? 0 : slot + 1;
            const int rslot = nslot + 1 == NS ? 0 : nslot + 1;     // slot of chunk c - 1 = of chunk c + 2
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int set = (i + PA) & 1;
                if (i + 1 < TM && !(MFTX_SABL & 8)) read_a(slot, i + 1, set ^ 1);
                if (i == BAR) {
                    if (!(MFTX_SABL & 4)) {
                        wait_vmcnt<0>();             // own pieces of chunk c + 1
                        block_barrier();
                    }
                    fetch_begin();
                }
                if (i >= BAR && i < BAR + TN && !(MFTX_SABL & 8)) read_b(nslot, i - BAR, PB ^ 1);
                if (i == TM - 1 && !(MFTX_SABL & 8)) read_a(nslot, 0, set ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < 3 * TN; ++m) {
                    const int j = m % TN, o = m / TN;
                    // (as assembly, accumulating in place in AGPRs: given the builtin and 512 registers the allocator shuttles
                    // half the accumulators between the two files on every trip.  Hazards by construction: operands come
                    // from ds_reads the compiler waits for; an accumulator is touched again 4 MFMAs later at the earliest.)
                    if (o == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fa16[set][0]), "v"(fb16[PB][j][0]));
                    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(accx[i][j]) : "v"(fa16[set][o]), "v"(fb16[PB][j][o]));
                    __builtin_amdgcn_sched_barrier(0);
                    if (m % 4 == 1 && i >= BAR && i < BAR + 4 && (i - BAR) * PPT + m / 4 < L && !(MFTX_SABL & 2)) {
                        fetch_piece(rslot, (i - BAR) * PPT + m / 4);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (i == BAR + 3) fetch_end();
            }
            slot = nslot;
        };
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        using PA1 = std::integral_constant<int, TM & 1>;
        int c = 0;
        for (; c + 1 < T; c += 2) {
            chunk(P0{}, P0{});
            chunk(PA1{}, P1{});
        }
        if (c < T) chunk(P0{}, P0{});
        wait_vmcnt<0>();
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");      // the last MFMAs' results, before the VALU reads them
    } else
    if constexpr (WS) {
        // Warp-specialised form of the loop below.  Producer waves: chunks 0 .. NS - 1 up front; then, per chunk c, wait for
        // their own pieces of chunk c + 1 (counted: NS - 2 younger chunks may fly), meet the consumers at the chunk's one
        // barrier -- which also certifies that every consumer has read the last of chunk c -- and refill that slot with
        // chunk c + NS.  Consumer waves: the loop below without a single DMA piece or vmcnt wait in it.
        constexpr int L = RA + RB;
        if (is_prod) {
#pragma unroll
            for (int i = 0; i < NS; ++i)
                if (i < T) fetch(i);
            if (T >= NS) wait_vmcnt<(NS - 1) * L>(); else wait_vmcnt<0>();
            block_barrier();
            int slot = 0;
            for (int c = 0; c + 1 < T; ++c) {
                if (c + NS - 1 < T) wait_vmcnt<(NS - 2) * L>(); else wait_vmcnt<0>();
                block_barrier();
                if (c + NS < T) fetch(slot);
                slot = slot + 1 == NS ? 0 : slot + 1;
            }
        } else {
            block_barrier();
            read_raw(0, 0, 0);
            ah[0] = __builtin_bit_cast(f16x8, ra[0][0][0]);
            al[0] = __builtin_bit_cast(f16x8, ra[0][0][1]);
            int slot = 0;
            auto chunk = [&](auto more_) {
                constexpr bool more = decltype(more_)::value;
                const int nslot = slot + 1 == NS ? 0 : slot + 1;
                read_raw(slot, 1, 1);
                group(0, 1, std::true_type{}, std::false_type{}, 0);
                if constexpr (more) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    block_barrier();
                    read_raw(nslot, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                group(1, 0, std::integral_constant<bool, more>{}, std::false_type{}, slot);
                slot = nslot;
            };
            for (int c = 0; c + 1 < T; ++c) chunk(std::true_type{});
            chunk(std::false_type{});
        }
    } else
    if constexpr (SPLIT) {
        // Ring of NS chunks, slot of chunk c = c % NS (run-time index: the K loop is not unrolled over the slots).
        // Per chunk: two 16-wide k groups; the raw fragments of the next group are read from LDS before the
        // current group is split and multiplied; ONE barrier per chunk, taken before the last group's MFMAs --
        // it certifies both that chunk c + 1 has landed (each lane waits for its own DMA pieces: counted vmcnt,
        // NS - 2 younger chunks may still fly) and that every wave has read the last of chunk c, whose slot
        // is then refilled with chunk c + NS.
        // (counted waits: every wave issues all RA + RB pieces of a chunk -- fetch_piece does, whatever BM % RPP)
        constexpr int L = RA + RB;
#pragma unroll
        for (int i = 0; i < NS; ++i)
            if (i < T) fetch(i);
        if (T >= NS) wait_vmcnt<(NS - 1) * L>(); else wait_vmcnt<0>();
        block_barrier();
        read_raw(0, 0, 0);
        if ((MFTX_SABL & 1) || PRE) {
            ah[0] = __builtin_bit_cast(f16x8, ra[0][0][0]);
            al[0] = __builtin_bit_cast(f16x8, ra[0][0][1]);
        } else {
            split8(ra[0][0][0], ra[0][0][1], k2048, ah[0], al[0]);      // the only split that no MFMA hides
        }
        int slot = 0;
#ifdef MFTX_TIMING_LOOP
        unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        unsigned tot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
#ifdef MFTX_TIMING
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt1));
#endif
        STAMP(6);
        int c = 0;
        auto chunk = [&](auto more_, auto refill_) {
            constexpr bool more = decltype(more_)::value, refill = decltype(refill_)::value && !(MFTX_SABL & 2);
            const int nslot = slot + 1 == NS ? 0 : slot + 1;
#ifdef MFTX_TIMING_LOOP
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ts[0]), "+s"(ts[1]), "+s"(ts[2]), "+s"(ts[3]), "+s"(ts[4]), "+s"(ts[5]), "+s"(ts[6]));
            if (c > 0) {
                tot[0] += (unsigned)(ts[1] - ts[0]); tot[1] += (unsigned)(ts[2] - ts[1]); tot[2] += (unsigned)(ts[3] - ts[2]);
                tot[3] += (unsigned)(ts[4] - ts[3]); tot[4] += (unsigned)(ts[5] - ts[4]); tot[5] += (unsigned)(ts[6] - ts[5]);
                tot[6] += 1;
            }
            ts[0] = ts[6];
#endif
            read_raw(slot, 1, 1);
            group(0, 1, std::true_type{}, std::false_type{}, 0);
            STAMP(1);
            if constexpr (more) {
                if (!(MFTX_SABL & 4)) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    STAMP(2);
                    if constexpr (decltype(refill_)::value || NS == 2) wait_vmcnt<(NS - 2) * L>();      // (steady part: chunks c + 1 .. c + NS - 1 are in flight)
                    else if (c + NS - 1 < T) wait_vmcnt<(NS - 2) * L>(); else wait_vmcnt<0>();
                    STAMP(3);
                    block_barrier();
                    STAMP(4);
                }
                read_raw(nslot, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // every wave has read the last of chunk c (the barrier above): its slot takes chunk c + NS
            if constexpr (refill) fetch_begin();
            STAMP(5);
            __builtin_amdgcn_sched_barrier(0);
            group(1, 0, std::integral_constant<bool, more>{}, std::integral_constant<bool, refill>{}, slot);
            if constexpr (refill) fetch_end();
            STAMP(6);
            slot = nslot;
        };
        for (; c + NS < T; ++c) chunk(std::true_type{}, std::true_type{});      // steady part: a chunk follows, the slot is refilled
        for (; c + 1 < T; ++c) chunk(std::true_type{}, std::false_type{});      // the last NS - 1 chunks before
        chunk(std::false_type{}, std::false_type{});                            // the last one
#ifdef MFTX_TIMING_LOOP
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 7; ++i) atomicAdd(&mftx_dbg[i], (unsigned long long)tot[i]);
        }
#endif
#ifdef MFTX_TIMING
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt2));
#endif
    } else {
    // prologue: chunks 0 and 1 in flight, chunk 0 landed, first fragments -> slot 0
    fetch(0);
    if (T > 1) {
        fetch(1);
        // chunk 0 landed, chunk 1 may fly -- counted per lane, so only when every wave issues all RA + RB loads
        if constexpr (BM % RPP == 0 && BN % RPP == 0) wait_vmcnt<RA + RB>(); else wait_vmcnt<0>();
    } else {
        wait_vmcnt<0>();
    }
    block_barrier();
    read_frags(0, 0, 0);
    int it = 0;
    for (; it + 3 < T; it += 2) {
        step(0, true, true);
        step(1, true, true);
    }
    // 1, 2 or 3 chunks left
    if (it + 2 < T) {
        step(0, true, true);
        step(1, true, false);
        step(0, false, false);
    } else if (it + 1 < T) {
        step(0, true, false);
        step(1, false, false);
    } else {
        step(0, false, false);
    }
    }
    if constexpr (SPLIT) {          // fold the cross terms in
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < NR; ++r) acc[i][j][r] += accx[i][j][r] * (1.f / 2048.f);
    }

    // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    // Two phases per 32x32 tile: every global read first (addend, z, h), then the arithmetic and the
    // stores.  Interleaved per element, the compiler must assume that a store aliases the next
    // element's loads and serialises 16 x (load - wait - store - wait): ~10 us per tile, fully
    // exposed when a CU has a single tile (one flow pair per GPU).
    // MT = 32: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), r < 16;  MT = 16: col = lane & 15, row = 4 (lane >> 4) + r, r < 4
    if constexpr (EPI == EPI_VOLUME) {
        static_assert(MT == 32 && TM == 1 && TN == 4 && WN == 1 && BN == 128, "volume tile: 32 queries x one super-block per wave");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        block_barrier();                      // every wave is done with the ring: it becomes epilogue staging
        if (!(MFTX_SABL & 32)) volume_epilogue(p, acc[0], smem + wid * 4096, lane, m0 + wm * 32, n0 / BN, bz);
#ifdef MFTX_TIMING
        {
            unsigned long long tt3;
            asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt3)::"memory");     // (stores drained: the epilogue's real length)
            if (lane == 0) {
                atomicAdd(&mftx_dbg[8], tt1 - tt0); atomicAdd(&mftx_dbg[9], tt2 - tt1); atomicAdd(&mftx_dbg[10], tt3 - tt2); atomicAdd(&mftx_dbg[11], 1ull);
            }
        }
#endif
        continue;
    }
    if constexpr (SPLIT || MT == 32) {      // (the fp32-MFMA kernels of 32 x 32 tiles too: same C/D layout)
        // Vectorised epilogue: each 32 x 32 accumulator tile goes through 4 KiB of the wave's own LDS (the ring is
        // free now) and comes back as four float4 per lane -- row t * 8 + (lane >> 3), columns 4 (lane & 7) .. + 3:
        // the addend / z / h reads and the stores become 16-byte accesses, a quarter of the instructions of the
        // C/D layout's 4-byte ones (the epilogue is issue-bound: 64 stores per 64 x 64 wave tile), same 128-byte runs
        // per row.  Two phases per tile as below: every global read first, then arithmetic and stores.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        block_barrier();                      // every wave is done with the ring: it becomes epilogue staging
        if (WS && is_prod) continue;          // (producer waves hold no accumulators: on to the next tile's top barrier)
        // (16-row MFMA tiles: the epilogue tile is one 16-row tile x the wave's 64 columns -- four passes of 4 rows x 16
        // float4; staging rows padded to 68 floats so that the four lane groups' C/D writes spread over all banks)
        constexpr int ETN = MT == 32 ? TM * TN : TM;            // epilogue tiles per wave
        constexpr int STS = MT == 32 ? 32 : 68, RS = MT == 32 ? 8 : 4;
        float *st = smem + wid * (MT == 32 ? 1024 : 16 * STS);
        const int c4 = MT == 32 ? lane & 7 : lane & 15, rq = MT == 32 ? lane >> 3 : lane >> 4;
        const bool pre_add = p.addend != nullptr && p.residual_mode == 0;
        const bool any_add = pre_add || p.residual_mode == 1;
        // (an output whose rows are not 16-byte aligned -- ldo = 126 -- is stored value by value)
        const bool vec_out = EPI == EPI_GRU_ZR || EPI == EPI_GRU_Q || ((p.ldo & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0);
        // software pipeline over the wave's TM x TN tiles: the global reads (addend, z, h) of tile k + 1 are issued
        // before tile k is staged, combined and stored -- per tile they are a full memory round trip, and a z|r
        // launch moves 73 MB through its epilogue (addend + h in, z + r h out) against 29 MB for a plain layer
        struct TileLoads { f32x4 add[4], a0[4], a1[4]; };
        auto tile_live = [&](int k) { return BMW == BM || MT != 32 || (k % TM) < tm_act; };      // (rows past BM belong to the next tile)
        // ---- straight-line form (the common case: N a multiple of 4, 16-byte aligned rows, operands below 2 GiB): every
        // global access is a raw buffer access whose offset is out of range where the branchy form below skips it (rows
        // past M: beyond the resource's extent by themselves; columns past N: the offset is made so) -- no divergent
        // branch around a memory instruction, so the compiler COUNTS what is in flight when it waits for tile k + 1's
        // loads.  Behind per-lane `continue`s it cannot, waits for everything (s_waitcnt vmcnt(0)) in front of every
        // row's stores, and each of the 16 row groups of a 64 x 64 wave tile sat out the round trip of the previous
        // group's stores: 27 of a z|r launch's 80 us (tile-phase stamps, tools/conv_phase_timing.py).
        const long long max_ld = EPI == EPI_GRU_ZR || EPI == EPI_GRU_Q ? (p.ld_hx > p.ld_hf ? p.ld_hx : p.ld_hf) : (p.ldo > p.ld_addend ? p.ldo : p.ld_addend);
        const bool has_add = pre_add || p.residual_mode == 1;
        // (N % 4 != 0 -- the motion encoder's 126 channels: the last column group is stored value by value behind one
        // wave-uniform branch per tile; with an addend as well, the branchy form)
        const bool n_tail = (p.N & 3) != 0 && EPI != EPI_GRU_ZR && EPI != EPI_GRU_Q;
        const bool straight = vec_out && ((p.N & 3) == 0 || (n_tail && !has_add)) && (long long)(p.M + BMW) * (max_ld > 256 ? max_ld : 256) * 4 < 0x7fffffffLL && !(MFTX_SABL & 2048);
        if (straight) {
            auto run = [&](auto osplit_, auto add_, auto tail_) {
                constexpr bool OS = decltype(osplit_)::value, ADD = decltype(add_)::value, TAIL = decltype(tail_)::value;
                const int nbt = p.N & ~3;                 // TAIL: first column of the ragged group
                auto mk = [&](const void *ptr, long long ld) {
                    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(ptr), 0, (unsigned)((long long)p.M * ld * 4), 0x00020000);
                };
                const __amdgpu_buffer_rsrc_t rBias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.bias ? p.bias : p.w), 0, p.bias ? (unsigned)p.N * 4u : 0u, 0x00020000);
                const __amdgpu_buffer_rsrc_t rAdd = mk(ADD ? p.addend : p.w, ADD ? p.ld_addend : 0);
                const __amdgpu_buffer_rsrc_t rOut = mk(out, p.ldo);
                const __amdgpu_buffer_rsrc_t rZ = mk(p.z ? p.z : out, 128), rRh = mk(p.rh ? p.rh : out, 128);
                const __amdgpu_buffer_rsrc_t rHf = mk(p.hf ? p.hf : out, p.ld_hf), rHx = mk(p.hx ? p.hx : out, p.ld_hx);
                auto tnb = [&](int k) { return MT == 32 ? n0 + wn * TN * 32 + (k / TM) * 32 + c4 * 4 : n0 + wn * TN * 16 + c4 * 4; };
                auto tmb = [&](int k) { return MT == 32 ? m0 + wm * TM * 32 + (k % TM) * 32 + rq : m0 + wm * TM * 16 + k * 16 + rq; };
                // byte offset of (row m, column col) in a map of row stride ld, out of range unless ok
                auto off = [&](int m, int ld, int col, bool ok) { return ok ? (unsigned)(m * ld + col) * 4u : OOB; };
                auto st128 = [&](const f32x4 &v, __amdgpu_buffer_rsrc_t r, unsigned o) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, o, 0, 0); };
                // four channels nb .. nb + 3 of a split-form row (row_bytes = its byte offset): high halves, low halves 16 bytes on
                // (tail > 0, wave-uniform: this tile holds the ragged group; its lane stores `valid` values as 2-byte pieces)
                auto st_split = [&](const f32x4 &v, __amdgpu_buffer_rsrc_t r, unsigned row_bytes, int nb, bool ok, bool tail = false, int valid = 0) {
                    unsigned h0, h1, l0, l1;
                    const float k2048s = 2048.f;
                    split_pair(v[0], v[1], k2048s, h0, l0);
                    split_pair(v[2], v[3], k2048s, h1, l1);
                    const unsigned o = ok ? row_bytes + (unsigned)split_row_offset(nb) : OOB;
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{h0, h1}, r, o, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{l0, l1}, r, o + 16u, 0, 0);       // (OOB + 16 stays out of range)
                    if (TAIL && tail) {
                        const unsigned hs[4] = {h0 & 0xffffu, h0 >> 16, h1 & 0xffffu, h1 >> 16}, ls[4] = {l0 & 0xffffu, l0 >> 16, l1 & 0xffffu, l1 >> 16};
#pragma unroll
                        for (int e = 0; e < 3; ++e) {
                            const unsigned oe = e < valid ? row_bytes + (unsigned)split_row_offset(nb) + 2u * e : OOB;
                            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)hs[e], r, oe, 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)ls[e], r, oe + 16u, 0, 0);
                        }
                    }
                };
                TileLoads ld[2];
                auto loads = [&](int k, TileLoads &L) {
                    const int nb = tnb(k), mb = tmb(k);
                    const bool okc = nb < p.N && tile_live(k);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int m = mb + t * RS;
                        if constexpr (ADD) L.add[t] = buf_load(rAdd, off(m, p.ld_addend, nb, okc));
                        else L.add[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                        L.a0[t] = L.a1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if constexpr (EPI == EPI_GRU_ZR) {
                            L.a1[t] = buf_load(rHf, off(m, p.ld_hf, nb - 128, okc && nb >= 128));
                        } else if constexpr (EPI == EPI_GRU_Q) {
                            L.a0[t] = buf_load(rZ, off(m, 128, nb, okc));
                            L.a1[t] = buf_load(rHf, off(m, p.ld_hf, nb, okc));
                        }
                    }
                };
                constexpr int NBIAS = MT == 32 ? TN : 1;
                f32x4 bias_c[NBIAS];
#pragma unroll
                for (int j = 0; j < NBIAS; ++j) {
                    const int nb = tnb(j * (MT == 32 ? TM : 1));
                    if constexpr (TAIL) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) bias_c[j][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rBias, nb + e < p.N ? (unsigned)(nb + e) * 4u : OOB, 0, 0));
                    } else {
                        bias_c[j] = buf_load(rBias, nb < p.N ? (unsigned)nb * 4u : OOB);
                    }
                }
                loads(0, ld[0]);
#pragma unroll
                for (int k = 0; k < ETN; ++k) {
                    if (k + 1 < ETN) loads(k + 1, ld[(k + 1) & 1]);
                    const TileLoads &L = ld[k & 1];
                    const int nb = tnb(k), mb = tmb(k);
                    const bool okc = (TAIL ? nb + 3 < p.N : nb < p.N) && tile_live(k);
                    // TAIL: does this tile hold the ragged column group (wave-uniform), and is it this lane's
                    const int tile_c0 = nb - c4 * 4;
                    const bool tail_here = TAIL && tile_c0 <= nbt && nbt < tile_c0 + (MT == 32 ? 32 : 64) && tile_live(k);
                    const int valid = TAIL && nb == nbt ? p.N - nbt : 0;
                    const f32x4 bias4 = bias_c[MT == 32 ? k / TM : 0];
                    if constexpr (MT == 32) {
                        float *w = st + (4 * (lane >> 5)) * 32 + (lane & 31);
#pragma unroll
                        for (int r = 0; r < 16; ++r) w[((r & 3) + 8 * (r >> 2)) * 32] = acc[k % TM][k / TM][r];
                    } else {
                        float *w = st + (4 * (lane >> 4)) * STS + (lane & 15);
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) w[r * STS + 16 * j] = acc[k][j][r];
                    }
                    f32x4 v[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = *reinterpret_cast<const f32x4 *>(st + (t * RS + rq) * STS + c4 * 4);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int m = mb + t * RS;
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float sv = v[t][e] + bias4[e];
                            if (ADD && p.residual_mode == 0) sv += L.add[t][e];
                            if constexpr (EPI == EPI_RELU) {
                                o[e] = relu_keep_nan(sv) * p.out_scale;
                            } else if constexpr (EPI == EPI_GRU_ZR) {
                                const float g = fast_sigmoid(sv);
                                o[e] = nb < 128 ? g : g * L.a1[t][e];
                            } else if constexpr (EPI == EPI_GRU_Q) {
                                o[e] = gru_blend(L.a0[t][e], L.a1[t][e], fast_tanh(sv));
                            } else {
                                float g = act_fn(sv, p.act) * p.out_scale;
                                if (ADD && p.residual_mode == 1) g = relu_keep_nan(g + L.add[t][e]);
                                o[e] = g;
                            }
                        }
                        if constexpr (EPI == EPI_GRU_ZR) {
                            st128(o, rZ, off(m, 128, nb, okc && nb < 128));                        // z: fp32, read by the q epilogue only
                            if constexpr (OS) st_split(o, rRh, (unsigned)m * 512u, nb - 128, okc && nb >= 128);      // r h: an A operand of the q GEMM
                            else st128(o, rRh, off(m, 128, nb - 128, okc && nb >= 128));
                        } else if constexpr (EPI == EPI_GRU_Q) {
                            st128(o, rHf, off(m, p.ld_hf, nb, okc));
                            if constexpr (OS) st_split(o, rHx, (unsigned)(m * p.ld_hx) * 4u, nb, okc);
                        } else if constexpr (OS) {
                            st_split(o, rOut, (unsigned)(m * p.ldo) * 4u, nb, okc, tail_here, valid);
                        } else {
                            st128(o, rOut, off(m, p.ldo, nb, okc));
                            if (TAIL && tail_here) {
#pragma unroll
                                for (int e = 0; e < 3; ++e)
                                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[e]), rOut, off(m, p.ldo, nb + e, e < valid), 0, 0);   // (not __builtin_bit_cast of a vector element: this compiler then stores element 0 every time)
                            }
                        }
                    }
                }
            };
            using Y = std::true_type;
            using N_ = std::false_type;
            bool done = false;
            // (split-form outputs exist in the split arithmetic only)
            if constexpr (EPI != EPI_GRU_ZR && EPI != EPI_GRU_Q) {
                if ((p.N & 3) != 0) {
                    if constexpr (SPLIT) { if (p.out_split) run(Y{}, N_{}, Y{}); else run(N_{}, N_{}, Y{}); }
                    else run(N_{}, N_{}, Y{});
                    done = true;
                }
            }
            if (done) {}
            else if (SPLIT && p.out_split) { if constexpr (SPLIT) { if (has_add) run(Y{}, Y{}, N_{}); else run(Y{}, N_{}, N_{}); } }
            else { if (has_add) run(N_{}, Y{}, N_{}); else run(N_{}, N_{}, N_{}); }
        } else {
        TileLoads ld[2];
        auto tile_nb = [&](int k) { return MT == 32 ? n0 + wn * TN * 32 + (k / TM) * 32 + c4 * 4 : n0 + wn * TN * 16 + c4 * 4; };   // first of this lane's four columns
        auto tile_mb = [&](int k) { return MT == 32 ? m0 + wm * TM * 32 + (k % TM) * 32 + rq : m0 + wm * TM * 16 + k * 16 + rq; };
        auto issue_loads = [&](int k, TileLoads &L) {
            if (!tile_live(k)) return;
            const int nb = tile_nb(k), mb = tile_mb(k);
            const bool full = nb + 3 < p.N;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const long long m = mb + t * RS;
                L.add[t] = L.a0[t] = L.a1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (m >= p.M || nb >= p.N) continue;
                if (full) {
                    if (any_add) L.add[t] = *reinterpret_cast<const f32x4 *>(p.addend + m * p.ld_addend + nb);
                    if constexpr (EPI == EPI_GRU_ZR) {
                        if (nb >= 128) L.a1[t] = *reinterpret_cast<const f32x4 *>(p.hf + m * p.ld_hf + (nb - 128));
                    } else if constexpr (EPI == EPI_GRU_Q) {
                        L.a0[t] = *reinterpret_cast<const f32x4 *>(p.z + m * 128 + nb);
                        L.a1[t] = *reinterpret_cast<const f32x4 *>(p.hf + m * p.ld_hf + nb);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (nb + e < p.N && any_add) L.add[t][e] = p.addend[m * p.ld_addend + nb + e];
                }
            }
        };
        issue_loads(0, ld[0]);
#pragma unroll
        for (int k = 0; k < ETN; ++k) {
            if (k + 1 < ETN) issue_loads(k + 1, ld[(k + 1) & 1]);
            if (!tile_live(k)) continue;
            const TileLoads &L = ld[k & 1];
            const int nb = tile_nb(k);
            const bool full = nb + 3 < p.N;
            f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
            if (p.bias != nullptr) {
                if (full) bias4 = *reinterpret_cast<const f32x4 *>(p.bias + nb);
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e) bias4[e] = nb + e < p.N ? p.bias[nb + e] : 0.f;
            }
            if constexpr (MT == 32) {
                float *w = st + (4 * (lane >> 5)) * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) w[((r & 3) + 8 * (r >> 2)) * 32] = acc[k % TM][k / TM][r];
            } else {
                float *w = st + (4 * (lane >> 4)) * STS + (lane & 15);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) w[r * STS + 16 * j] = acc[k][j][r];
            }
            const int mb = tile_mb(k);
            f32x4 v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = *reinterpret_cast<const f32x4 *>(st + (t * RS + rq) * STS + c4 * 4);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const long long m = mb + t * RS;
                if (m >= p.M || nb >= p.N) continue;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float sv = v[t][e] + bias4[e];
                    if (pre_add) sv += L.add[t][e];
                    if constexpr (EPI == EPI_RELU) {
                        o[e] = relu_keep_nan(sv) * p.out_scale;
                    } else if constexpr (EPI == EPI_GRU_ZR) {   // [z | r] gates; r is folded into r*h
                        const float g = fast_sigmoid(sv);
                        o[e] = nb < 128 ? g : g * L.a1[t][e];
                    } else if constexpr (EPI == EPI_GRU_Q) {    // candidate q, h <- (1-z) h + z q
                        o[e] = gru_blend(L.a0[t][e], L.a1[t][e], fast_tanh(sv));
                    } else {
                        float g = act_fn(sv, p.act) * p.out_scale;
                        if (p.residual_mode == 1) g = relu_keep_nan(g + L.add[t][e]);
                        o[e] = g;
                    }
                }
                if constexpr (EPI == EPI_GRU_ZR) {
                    if (nb < 128) *reinterpret_cast<f32x4 *>(p.z + m * 128 + nb) = o;          // z: fp32, read by the q epilogue only
                    else if (p.out_split) store_split4(p.rh + m * 128, nb - 128, o);           // r h: an A operand of the q GEMM
                    else *reinterpret_cast<f32x4 *>(p.rh + m * 128 + (nb - 128)) = o;
                    continue;
                }
                if constexpr (EPI == EPI_GRU_Q) {
                    *reinterpret_cast<f32x4 *>(p.hf + m * p.ld_hf + nb) = o;
                    if (p.out_split) store_split4(p.hx + m * p.ld_hx, nb, o);
                    continue;
                }
                if (p.out_split) {
                    store_split4(out + m * p.ldo, nb, o, full ? 4 : p.N - nb);
                    continue;
                }
                float *dst = out + m * p.ldo + nb;
                if (full && vec_out) *reinterpret_cast<f32x4 *>(dst) = o;
                else
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (nb + e < p.N) dst[e] = o[e];
            }
        }
        }   // branchy form
#ifdef MFTX_TIMING
        if constexpr (MT == 32) {
            unsigned long long tt3;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt3)::"memory");      // (stores issued, not drained)
            if (lane == 0) {
                atomicAdd(&mftx_dbg[8], tt1 - tt0); atomicAdd(&mftx_dbg[9], tt2 - tt1); atomicAdd(&mftx_dbg[10], tt3 - tt2); atomicAdd(&mftx_dbg[11], 1ull);
            }
        }
#endif
        continue;
    }
    const int col_l = lane & (MT - 1);
    const int row_h = MT == 32 ? 4 * (lane >> 5) : 4 * (lane >> 4);
    auto row_of = [](int r) { return MT == 32 ? (r & 3) + 8 * (r >> 2) : r; };
    const bool pre_add = p.addend != nullptr && p.residual_mode == 0;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * MT + j * MT + col_l;
        const bool n_ok = n < p.N;
        const float bias = (p.bias != nullptr && n_ok) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float add[NR], aux0[NR], aux1[NR];       // addend; z / h (GRU)
            const int mb = m0 + wm * TM * MT + i * MT + row_h;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int m = mb + row_of(r);
                add[r] = aux0[r] = aux1[r] = 0.f;
                if (!n_ok || m >= p.M) continue;
                if (pre_add || p.residual_mode == 1) add[r] = p.addend[(long long)m * p.ld_addend + n];
                if constexpr (EPI == EPI_GRU_ZR) {
                    if (n >= 128) aux1[r] = p.hx[(long long)m * p.ld_hx + (n - 128)];
                } else if constexpr (EPI == EPI_GRU_Q) {
                    aux0[r] = p.z[(long long)m * 128 + n];
                    aux1[r] = p.hx[(long long)m * p.ld_hx + n];
                }
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const long long m = mb + row_of(r);
                if (!n_ok || m >= p.M) continue;
                float s = acc[i][j][r] + bias;
                if (pre_add) s += add[r];
                if constexpr (EPI == EPI_RELU) {
                    out[m * p.ldo + n] = relu_keep_nan(s) * p.out_scale;
                } else if constexpr (EPI == EPI_GRU_ZR) {   // [z | r] gates; r is folded into r*h
                    const float v = fast_sigmoid(s);
                    if (n < 128) p.z[m * 128 + n] = v;
                    else p.rh[m * 128 + (n - 128)] = v * aux1[r];
                } else if constexpr (EPI == EPI_GRU_Q) {    // candidate q, h <- (1-z) h + z q
                    const float v = fast_tanh(s);
                    p.hx[m * p.ld_hx + n] = gru_blend(aux0[r], aux1[r], v);
                } else {
                    float v = act_fn(s, p.act) * p.out_scale;
                    if (p.residual_mode == 1) v = relu_keep_nan(v + add[r]);
                    out[m * p.ldo + n] = v;
                }
            }
        }
    }
    }  // persistent tile loop
}

__device__ __forceinline__ int n_virtual_tiles(const ConvArgs &p, int BM, int BN) {
    return 8 * (((p.M + BM - 1) / BM + 7) / 8) * ((p.N + BN - 1) / BN) * p.batch;
}

template <int BM, int BN, int WM, int WN, int EPI, int MT = 32, int AR = AR_F32, int NS = 2, int WS = 0>
__global__ __launch_bounds__(64 * WM * WN * (1 + WS), WS ? 2 * WM * WN / 4 : AR != AR_F32 ? min_waves_split((BM / WM / MT) * (BN / WN / MT), WM * WN, MT) : min_waves((BM / WM / MT) * (BN / WN / MT), EPI))
void conv_gemm_kernel(ConvArgs p) {
    conv_gemm_body<BM, BN, WM, WN, EPI, MT, AR, NS, WS>(p, blockIdx.x);
}

// Two independent convolutions in one launch (the two branches of the motion encoder, convc2 and convf2):
// the virtual tiles of b follow those of a in the same round-robin, so the pair fills the chip as one
// problem -- at one flow pair per GPU 192 + 64 tiles for 256 CUs instead of two half-empty launches, at
// seven pairs 1344 + 448 = 7 per CU instead of 2.6 and 1.75 rounds.  Same tiles, same results.
template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(64 * WM * WN, min_waves((BM / WM / 32) * (BN / WN / 32)))
void conv_gemm_pair_kernel(ConvArgs a, ConvArgs b) {
    conv_gemm_body<BM, BN, WM, WN, EPI, 32>(a, blockIdx.x);
    const int na = n_virtual_tiles(a, BM, BN);           // a multiple of 8: b keeps its XCD mapping
    int vt = (int)blockIdx.x - na % (int)gridDim.x;
    if (vt < 0) vt += gridDim.x;
    conv_gemm_body<BM, BN, WM, WN, EPI, 32>(b, vt);
}

static int num_cus() {
    static const int n = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                cus = prop.multiProcessorCount;
        }
        return cus;
    }();
    return n;
}

template <int BM, int BN, int WM, int WN, int EPI, int MT = 32, int AR = AR_F32, int NS = 2, int WS = 0>
static int launch_cfg(const ConvArgs &a, int batch, hipStream_t s, ProfCat cat, double work = -1.0) {
    constexpr int RPP = 8 * WM * WN;                  // ring slots hold whole staging passes
    constexpr size_t lds = (size_t)NS * ((BM + RPP - 1) / RPP + (BN + RPP - 1) / RPP) * RPP * LDK * sizeof(float);
    static bool attr_set = false;
    auto kern = conv_gemm_kernel<BM, BN, WM, WN, EPI, MT, AR, NS, WS>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    ConvArgs args = a;
    args.batch = batch;
    // resident workgroups per CU: LDS-bound (160 KiB per CU), at most MFTX_CONV_RESIDENT_WAVES waves
    static const int max_waves = tune_env("MFTX_CONV_RESIDENT_WAVES", 16);
    const int max_res = max_waves / (WM * WN * (1 + WS));
    const int resident = (160 * 1024) / (int)lds < max_res ? (160 * 1024) / (int)lds : max_res;
    const long long n_virtual = 8ll * cdiv(cdiv(a.M, BM), 8) * cdiv(a.N, BN) * batch;
    static const bool one_tile_per_wg = tune_env("MFTX_CONV_NONPERSISTENT", 0) != 0;   // tuning: let the dispatcher interleave kernels of two streams
    const long long slots = one_tile_per_wg ? n_virtual : (long long)num_cus() * resident;
    dim3 grid((unsigned)(n_virtual < slots ? n_virtual : slots));
    // algorithmic flops: real (unpadded) reduction length
    ProfScope prof(cat, s, work >= 0 ? work : 2.0 * a.M * a.N * (double)(a.kh * a.kw) * (a.c0 + a.c1) * batch);
    hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN * (1 + WS)), lds, s, args);
    return check_launch("conv_gemm");
}

// tile shapes: 0 = 128x128, 1 = 128x64, 2 = 64x64, 3 = 128x32.  LDS rings of 3 and 4 chunks (counted
// vmcnt) were measured twice and do not pay at either M = 28672 or M = 4096 (tools/bench_conv.py):
// the ring is fixed at two chunks, which lets the K loop be unrolled over the slots.
// (One instantiation per epilogue, each with ~19 kernels: compiled as four translation units beside this one -- the same
// file with -DMFTX_CONV_PART=<epilogue>, see the Makefile -- so that the build parallelises; -DMFTX_CONV_SINGLE_TU keeps
// everything here, for the tuning builds of tools/build_ablations.sh.)
template <int EPI>
int launch_tile(int tile, const ConvArgs &a, int batch, hipStream_t s, ProfCat cat) {
    if constexpr (EPI == EPI_GRU_Q) {
        // the candidate epilogue keeps z, h and r * h fragments live beside the accumulators: on three shapes that does not fit
        // 256 VGPRs (17 / 6 / 10 spilled registers).  They are reachable only with the fused GRU pass switched off (or a forced
        // tile) and every tile shape sums K in the same order (bit-identical results), so they run on their no-spill neighbours.
        if (a.arith == AR_SPLIT && a.a_pre && tile == 11) tile = 6;
        if (a.arith != AR_SPLIT && (tile == 1 || tile == 4)) tile = 2;
    }
    if (a.arith == AR_SPLIT && a.a_pre) {              // A already in split form: the production tiles only
        switch (tile) {
            case 0: return launch_cfg<128, 128, 2, 2, EPI, 32, AR_PRESPLIT, 2>(a, batch, s, cat);
            case 6: return launch_cfg<128, 128, 4, 2, EPI, 32, AR_PRESPLIT, 3>(a, batch, s, cat);       // (ring of three: 125.1 -> 126.2 frames/s against four)
            case 10: return launch_cfg<128, 256, 2, 4, EPI, 32, AR_PRESPLIT, 3>(a, batch, s, cat);
            case 14: return launch_cfg<128, 192, 4, 2, EPI, 32, AR_PRESPLIT, 3>(a, batch, s, cat);   // eight waves of 32 x 96: N = 192 without a half-empty column tile
            case 11: if constexpr (EPI != EPI_GRU_Q) return launch_cfg<128, 128, 2, 2, EPI, 32, AR_PRESPLIT, 4, 1>(a, batch, s, cat); else break;   // warp-specialised: four 64 x 64 consumer waves + four producer waves, ring of four chunks (128 KiB)
#ifdef MFTX_EXPERIMENTAL_TILES       // measurement-only shapes: bit-identical, slower
            case 15: return launch_cfg<224, 128, 2, 4, EPI, 32, AR_PRESPLIT, 3>(a, batch, s, cat);   // four waves of 128 x 32 over four of 96 x 32: 7 x 4096 cells = 128 x 224
            case 13: return launch_cfg<112, 256, 1, 4, EPI, 16, AR_PRESPLIT, 3>(a, batch, s, cat);  // four waves of 112 x 64 on 16-row MFMAs: 7 x 4096 cells = 256 tiles
#endif
            default: return launch_cfg<64, 64, 2, 2, EPI, 32, AR_PRESPLIT, 3>(a, batch, s, cat);
        }
        return fail(MFTX_E_STATE, "conv2d: tile %d has no instance for this epilogue", tile);
    }
    if (a.arith == AR_SPLIT) {
        switch (tile) {
            case 0: return launch_cfg<128, 128, 2, 2, EPI, 32, AR_SPLIT, 2>(a, batch, s, cat);   // two workgroups of four 64 x 64 waves per CU
            case 6: return launch_cfg<128, 128, 4, 2, EPI, 32, AR_SPLIT, 4>(a, batch, s, cat);   // eight waves of 32 x 64
            case 10: return launch_cfg<128, 256, 2, 4, EPI, 32, AR_SPLIT, 3>(a, batch, s, cat);  // eight waves of 64 x 64: the A tile is staged once for 256 output channels; ring of three chunks (144 KiB): standalone layers lose a microsecond to it, the engine -- inputs fresh from the previous kernel, longer latencies -- gains 2.3 % (123.9 -> 126.7 frames/s)
            case 14: return launch_cfg<128, 192, 4, 2, EPI, 32, AR_SPLIT, 3>(a, batch, s, cat);
#ifdef MFTX_EXPERIMENTAL_TILES
            case 15: return launch_cfg<224, 128, 2, 4, EPI, 32, AR_SPLIT, 3>(a, batch, s, cat);
#endif
            default: return launch_cfg<64, 64, 2, 2, EPI, 32, AR_SPLIT, 3>(a, batch, s, cat);        // 9
        }
    }
    switch (tile) {
        case 0: return launch_cfg<128, 128, 2, 2, EPI>(a, batch, s, cat);
        case 1: if constexpr (EPI != EPI_GRU_Q) return launch_cfg<128, 64, 2, 2, EPI>(a, batch, s, cat); else break;
        case 2: return launch_cfg<64, 64, 2, 2, EPI>(a, batch, s, cat);
        case 4: if constexpr (EPI != EPI_GRU_Q) return launch_cfg<64, 128, 2, 2, EPI>(a, batch, s, cat); else break;       // a wave owns 32 x 64: the A tile is gathered once for N = 128
        case 5: return launch_cfg<32, 32, 2, 2, EPI, 16>(a, batch, s, cat);    // 4 waves of 16x16: four times the workgroups of 64x64
        default: return launch_cfg<128, 32, 4, 1, EPI>(a, batch, s, cat);
    }
    return fail(MFTX_E_STATE, "conv2d: tile %d has no instance for this epilogue", tile);
}

#if defined(MFTX_CONV_PART)
template int launch_tile<MFTX_CONV_PART>(int, const ConvArgs &, int, hipStream_t, ProfCat);
#else
#if !defined(MFTX_CONV_SINGLE_TU)
extern template int launch_tile<EPI_GENERIC>(int, const ConvArgs &, int, hipStream_t, ProfCat);
extern template int launch_tile<EPI_RELU>(int, const ConvArgs &, int, hipStream_t, ProfCat);
extern template int launch_tile<EPI_GRU_ZR>(int, const ConvArgs &, int, hipStream_t, ProfCat);
extern template int launch_tile<EPI_GRU_Q>(int, const ConvArgs &, int, hipStream_t, ProfCat);
#endif

static int pick_tile(const ConvArgs &a, int batch, int forced_arg) {
    // forced: mftx_conv2d_tile (tests, micro-benchmarks); tuning builds also read MFTX_CONV_TILE
    static const int forced_env = tune_env("MFTX_CONV_TILE", -1);
    const int forced = forced_arg >= 0 ? forced_arg : forced_env;
    if (forced >= 0 && forced <= 15) return forced;
    if (a.arith == AR_SPLIT) {
        // Measured (tools/bench_conv.py --arith 1, M = 7 x 4096).  The staging path (global -> LDS) is what limits
        // these kernels, so the biggest tile that still fills the chip wins:
        //   10: 128 x 256, eight 64 x 64 waves (A staged once for 256 output channels)   N % 256 == 0, >= 3/4 tile per CU
        //    0: 128 x 128, four 64 x 64 waves, two workgroups per CU                     >= 1.5 tiles per CU
        //    6: 128 x 128, eight 32 x 64 waves, one workgroup per CU                     3/4 .. 1 tile per CU
        //    9:  64 x  64, four 32 x 32 waves, ring of three chunks, three workgroups per CU    small N, small M
        const long long cus = num_cus();
        const long long t128 = (long long)cdiv(a.M, 128) * cdiv(a.N, 128);
        if (a.N <= 64) return 9;
        //   14: 128 x 192, eight 32 x 96 waves, one workgroup per CU: N = 192 (convc2) without the half-empty second column tile
        //       of the 128-wide shapes -- a quarter of their matrix work (measured at M = 7 x 4096: 110.5 -> 88.7 us)
        static const bool no14 = tune_env("MFTX_CONV_NO14", 0) != 0;
        // (one round of workgroups only: N = 576 at seven pairs, 672 tiles in three rounds, is faster on 128-wide tiles: 45.3 vs 38.5 us)
        if (!no14 && a.N % 192 == 0 && a.N % 128 != 0 && (long long)cdiv(a.M, 128) * (a.N / 192) * 2 >= cus && (long long)cdiv(a.M, 128) * (a.N / 192) <= cus) return 14;
        static const bool try13 = tune_env("MFTX_CONV_TILE13", 0) != 0;     // tuning: the 112-row tile where it fills the chip in one round
        if (try13 && a.a_pre && a.N % 256 == 0 && (long long)cdiv(a.M, 112) * (a.N / 256) <= cus && (long long)cdiv(a.M, 112) * (a.N / 256) * 8 >= cus * 7) return 13;
        if (a.N % 256 == 0 && (long long)cdiv(a.M, 128) * (a.N / 256) * 4 >= cus * 3) return 10;
        if (t128 * 2 >= cus * 3) return 0;
        //   11: 128 x 128 warp-specialised (four 64 x 64 consumer waves + four producer waves), A pre-split: the same one round
        //       as tile 6, measured at M = 7 x 4096: conv 3x3 256->126 64.0 -> 57.3 us, q gates 40.9 / 36.7 -> 37.4 / 35.6 us
        if (t128 * 4 >= cus * 3 && t128 <= cus) return a.a_pre ? 11 : 6;     // one round only: five pairs, N = 256 (320 tiles) would take two
        return 9;
    }
    // Measured on MI355X (tools/bench_conv.py, M = 7 x 4096 and 4096): the 64x64
    // tile (4 workgroups per CU) wins or ties on every layer -- 7 x 2^k rows tile
    // the 256 CUs more evenly in small tiles than bigger tiles save on operand
    // re-reads -- except for N = 192 (convc2), where 128x64 avoids a half-empty
    // column tile.  N <= 32 (only reached when the small-N VALU kernel does not
    // apply) gets the 128x32 tile.  A two-wave 64x32 tile (twice the tiles) was
    // measured and loses 20 %.
    (void)batch;
    if (a.N <= 32) return 3;
    if (a.N > 128 && a.N % 128 == 64 && a.M >= 16384) return 1;   // at small M the extra tiles of 64x64 win
    // Very small M x N (one flow pair per GPU, N = 64: 64 tiles of 64x64 for 256 CUs): 32x32 tiles of four
    // 16x16-MFMA waves spread the same outputs over four times as many workgroups -- same results, bit for
    // bit.  Measured at M = 4096 (tools/bench_conv.py): N = 64 25.5 -> 18.9 us; already at N = 128 the
    // 16x16 form loses (27.5 -> 29 us; a 32x64 tile of 8 waves 47.6 -> 81 us at N = 192): per MFMA cycle it
    // needs four times the LDS fragment traffic and twice the LDS-DMA pieces of the 32x32 form.
    static const bool small_off = tune_env("MFTX_CONV_NO16", 0) != 0;
    const long long t64 = (long long)cdiv(a.M, 64) * cdiv(a.N, 64);
    if (!small_off && t64 * 4 <= num_cus()) return 5;
    return 2;
}

static int dispatch(const ConvArgs &a, int epi, int batch, hipStream_t s, ProfCat cat, int forced_tile = -1) {
    const int tile = pick_tile(a, batch, forced_tile);
#ifndef MFTX_EXPERIMENTAL_TILES
    if (tile == 13 || tile == 15) return fail(MFTX_E_ARG, "conv2d: tile %d is a measurement-only shape (build with -DMFTX_EXPERIMENTAL_TILES)", tile);
#endif
    switch (epi) {
        case EPI_RELU: return launch_tile<EPI_RELU>(tile, a, batch, s, cat);
        case EPI_GRU_ZR: return launch_tile<EPI_GRU_ZR>(tile, a, batch, s, cat);
        case EPI_GRU_Q: return launch_tile<EPI_GRU_Q>(tile, a, batch, s, cat);
        default: return launch_tile<EPI_GENERIC>(tile, a, batch, s, cat);
    }
}

static int validate(const mftx_conv_desc &d) {
    if (!d.a0 || !d.wpk || !d.out) return fail(MFTX_E_ARG, "conv2d: null pointer");
    if (d.P <= 0 || d.h <= 0 || d.w <= 0 || d.N <= 0 || d.c0 <= 0 || d.c1 < 0)
        return fail(MFTX_E_ARG, "conv2d: bad sizes");
    if (d.kh < 1 || d.kw < 1 || !(d.kh & 1)) return fail(MFTX_E_ARG, "conv2d: kernel height must be odd");
    if (d.stride < 0 || d.stride > 4 || d.hin < 0 || d.win < 0) return fail(MFTX_E_ARG, "conv2d: bad stride / input grid");
    if (d.residual_mode != 0 && (d.residual_mode != 1 || !d.addend)) return fail(MFTX_E_ARG, "conv2d: bad residual mode");
    if (d.c1 > 0 && (!d.a1 || d.c0 % BK)) return fail(MFTX_E_ARG, "conv2d: segment 0 must be a multiple of 32 channels");
    if ((d.c0 + d.c1) % 4 || d.c0 % 4) return fail(MFTX_E_ARG, "conv2d: channel counts must be multiples of 4");
    if (d.lda0 % 4 || (d.c1 > 0 && d.lda1 % 4) || !aligned16(d.a0) || (d.c1 > 0 && !aligned16(d.a1)) || !aligned16(d.wpk))
        return fail(MFTX_E_ALIGN, "conv2d: operands must be 16-byte aligned");
    if (d.act < 0 || d.act > 3) return fail(MFTX_E_ARG, "conv2d: bad activation");
    if (d.arith != AR_F32 && d.arith != AR_SPLIT) return fail(MFTX_E_ARG, "conv2d: bad arithmetic");
    if ((d.a_split || d.out_split) && d.arith != AR_SPLIT) return fail(MFTX_E_ARG, "conv2d: split-form operands belong to the split arithmetic");
    if (d.a_split && (d.c0 % 8 || d.c1 % 8 || d.lda0 % 8 || (d.c1 > 0 && d.lda1 % 8) || (reinterpret_cast<uintptr_t>(d.a0) & 31) || (d.c1 > 0 && (reinterpret_cast<uintptr_t>(d.a1) & 31))))
        return fail(MFTX_E_ALIGN, "conv2d: a split-form A operand has channel counts and row strides in multiples of 8 and 32-byte aligned rows");
    if (d.out_split && (d.ldo % 8 || (reinterpret_cast<uintptr_t>(d.out) & 31)))
        return fail(MFTX_E_ALIGN, "conv2d: a split-form output has a row stride in multiples of 8 and 32-byte aligned rows");
    if (d.arith == AR_SPLIT && ((d.bias && !aligned16(d.bias)) || (d.addend && (d.ld_addend % 4 || !aligned16(d.addend)))))
        return fail(MFTX_E_ALIGN, "conv2d: split arithmetic reads 16-byte pieces: bias and addend must be 16-byte aligned, the addend's row stride a multiple of 4");
    const long long M = (long long)d.P * d.h * d.w;
    const long long Min = (long long)d.P * (d.hin ? d.hin : d.h) * (d.win ? d.win : d.w);
    const long long lim = 0x7fffffffLL;      // buffer offsets are 32-bit, bit 31 marks "out of range"
    if (M > lim || Min * d.lda0 * 4 > lim || (d.c1 > 0 && Min * d.lda1 * 4 > lim))
        return fail(MFTX_E_ARG, "conv2d: activation operand exceeds 2 GiB");
    const long long wbytes = (long long)round_up(d.N, 128) * d.kh * d.kw * round_up(d.c0 + d.c1, BK) * 4;
    if (wbytes > lim) return fail(MFTX_E_ARG, "conv2d: weight operand exceeds 2 GiB");
    return 0;
}

static ConvArgs to_args(const mftx_conv_desc &d) {
    ConvArgs a{};
    a.a0 = d.a0; a.a1 = d.a1; a.lda0 = d.lda0; a.lda1 = d.lda1; a.c0 = d.c0; a.c1 = d.c1;
    a.w = d.wpk; a.bias = d.bias; a.out = d.out; a.ldo = d.ldo;
    a.M = d.P * d.h * d.w; a.N = d.N; a.h = d.h; a.wd = d.w; a.kh = d.kh; a.kw = d.kw;
    a.cin_pad = round_up(d.c0 + d.c1, BK);
    a.w_rows = round_up(d.N, 128);
    a.act = d.act; a.out_scale = d.out_scale;
    a.arith = d.arith;
    a.a_pre = d.a_split; a.out_split = d.out_split;
    a.hf = nullptr; a.ld_hf = 0;
    a.addend = d.addend; a.ld_addend = d.ld_addend; a.residual_mode = d.residual_mode;
    a.stride = d.stride ? d.stride : 1;
    a.hin = d.hin ? d.hin : d.h; a.win = d.win ? d.win : d.w;
    a.pad_y = d.pad_y < 0 ? 0 : (d.pad_y ? d.pad_y : d.kh / 2);   // 0 = "same" default, -1 = no padding
    a.pad_x = d.pad_x < 0 ? 0 : (d.pad_x ? d.pad_x : d.kw / 2);
    // extents: the last input cell's row ends at (Min-1)*lda + c
    const long long Min = (long long)d.P * a.hin * a.win;
    a.a0_bytes = (unsigned)(((Min - 1) * d.lda0 + d.c0) * 4);
    a.a1_bytes = d.c1 > 0 ? (unsigned)(((Min - 1) * d.lda1 + d.c1) * 4) : 0u;
    a.w_bytes = (unsigned)((long long)a.w_rows * d.kh * d.kw * a.cin_pad * 4);
    return a;
}

int launch_conv(const mftx_conv_desc &d, hipStream_t s, int tile) {
    if (int e = validate(d)) return e;
    if (d.addend == nullptr && d.stride <= 1 && d.hin == 0 && conv_small_applicable(d)) {
        if (d.arith != AR_F32) return fail(MFTX_E_ARG, "conv2d: split arithmetic is for the matrix path (N > 4); this layer runs on the VALU kernel, with fp32 weights");
        return launch_conv_small(d, s);
    }   // N <= 4: VALU kernel, no MFMA padding waste
    const bool relu = d.act == 1 && d.residual_mode == 0;   // the residual tail lives in the generic epilogue
    return dispatch(to_args(d), relu ? EPI_RELU : EPI_GENERIC, 1, s, PC_CONV_GEMM, tile);
}

int launch_conv_pair(const mftx_conv_desc &da, const mftx_conv_desc &db, hipStream_t s) {
    if (int e = validate(da)) return e;
    if (int e = validate(db)) return e;
    if (da.act != 1 || db.act != 1 || da.residual_mode || db.residual_mode || da.N <= 32 || db.N <= 32 || da.arith != AR_F32 || db.arith != AR_F32)
        return fail(MFTX_E_ARG, "conv pair: two fp32-MFMA ReLU convolutions with more than 32 output channels");
    ConvArgs a = to_args(da), b = to_args(db);
    a.batch = b.batch = 1;
    constexpr int BM = 64, BN = 64;
    constexpr size_t lds = (size_t)2 * (BM + BN) * LDK * sizeof(float);
    auto kern = conv_gemm_pair_kernel<BM, BN, 2, 2, EPI_RELU>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const long long n_virtual = 8ll * cdiv(cdiv(a.M, BM), 8) * cdiv(a.N, BN) + 8ll * cdiv(cdiv(b.M, BM), 8) * cdiv(b.N, BN);
    static const bool one_tile_per_wg = tune_env("MFTX_CONV_NONPERSISTENT", 0) != 0;
    const long long slots = one_tile_per_wg ? n_virtual : (long long)num_cus() * 4;
    dim3 grid((unsigned)(n_virtual < slots ? n_virtual : slots));
    ProfScope prof(PC_CONV_GEMM, s, 2.0 * a.M * a.N * (double)(a.kh * a.kw) * (a.c0 + a.c1) +
                                        2.0 * b.M * b.N * (double)(b.kh * b.kw) * (b.c0 + b.c1));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a, b);
    return check_launch("conv_gemm pair");
}

int launch_conv_gru(const mftx_conv_desc &d, const GruEpilogue &g, hipStream_t s) {
    if (int e = validate(d)) return e;
    ConvArgs a = to_args(d);
    a.hx = g.hx; a.ld_hx = g.ld_hx; a.z = g.z; a.rh = g.rh;
    a.hf = g.hf ? g.hf : g.hx; a.ld_hf = g.hf ? g.ld_hf : g.ld_hx;
    if ((g.mode == 1 && d.N != 256) || (g.mode == 2 && d.N != 128)) return fail(MFTX_E_ARG, "gru epilogue: bad N");
    return dispatch(a, g.mode == 1 ? EPI_GRU_ZR : EPI_GRU_Q, 1, s, PC_CONV_GEMM);
}

// Packed fp32 weights [rows][taps * cin_pad] -> the split form the AR_SPLIT kernels stream: the same index space
// and size, every 8 consecutive k (32 bytes) holding [hi x 8 | lo x 8] as fp16, hi = fp16(w), lo = fp16((w - hi) * 2048)
__global__ void split_weights_kernel(const float *__restrict__ in, uint4 *__restrict__ out, long long n8) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    _Float16 hi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = in[i * 8 + e];
        hi[e] = (_Float16)x;
        lo[e] = (_Float16)((x - (float)hi[e]) * 2048.f);
    }
    uint4 h, l;
    __builtin_memcpy(&h, hi, 16);
    __builtin_memcpy(&l, lo, 16);
    out[2 * i] = h;
    out[2 * i + 1] = l;
}

#ifdef MFTX_TIMING
extern "C" int mftx_debug_timing(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(mftx_dbg), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(mftx_dbg), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

int launch_split_weights(const float *wpk, void *out, long long n, hipStream_t s) {
    if (n <= 0 || n % 8) return fail(MFTX_E_ARG, "split_weights: the packed weight size must be a positive multiple of 8 floats");
    if (!aligned16(wpk) || !aligned16(out)) return fail(MFTX_E_ALIGN, "split_weights: operands must be 16-byte aligned");
    const long long n8 = n / 8;
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)cdiv(n8, 256)), dim3(256), 0, s, wpk, reinterpret_cast<uint4 *>(out), n8);
    return check_launch("split_weights");
}

// lvl[0][p][i][.] = <f1[p][i][:], f2[p][j][:]> / sqrt(C) over all target cells j (blocked layout), and the
// three pooled levels, in one launch (core/corr.py:14-28, 53-69)
// f2_split (optional): scratch of P * h * w * C floats -> the volume GEMM runs in split arithmetic (f2 is split into it first)
int launch_corr_pyramid(const float *f1, const float *f2, int P, int C, int h, int w, float *const lvl[4], hipStream_t s, float *f2_split, int tile_resident) {
    const int N = h * w;
    if ((long long)N * C * 4 > 0x7fffffffLL) return fail(MFTX_E_ARG, "corr_pyramid: feature map exceeds 2 GiB");
    const PyramidLayout L = pyramid_layout(h, w);
    ConvArgs a{};
    a.a0 = f1; a.lda0 = C; a.c0 = C; a.c1 = 0; a.a1 = nullptr; a.lda1 = 0;
    a.w = f2; a.bias = nullptr; a.out = lvl[0]; a.ldo = 0;
    a.M = N; a.N = L.sbh * L.sbw * 128; a.h = 1; a.wd = N; a.kh = 1; a.kw = 1; a.cin_pad = C;
    a.stride = 1; a.hin = 1; a.win = N; a.pad_y = 0; a.pad_x = 0;
    a.w_rows = N;
    a.act = 0; a.out_scale = 1.0f / sqrtf((float)C);
    a.a_bstride = (long long)N * C; a.w_bstride = (long long)N * C; a.o_bstride = 0;   // the epilogue addresses by batch itself
    a.a0_bytes = (unsigned)((long long)N * C * 4); a.a1_bytes = 0; a.w_bytes = a.a0_bytes;
    a.vh = h; a.vw = w; a.sbw = L.sbw; a.wb0 = L.wb[0];
    a.lvl1 = lvl[1]; a.lvl2 = lvl[2]; a.lvl3 = lvl[3];
    a.s0 = L.stride[0]; a.s1 = L.stride[1]; a.s2 = L.stride[2]; a.s3 = L.stride[3];
    // one wave = 32 queries x one super-block (128 columns): 128 x 128 tiles of four waves stacked along M
    if (f2_split != nullptr) {
        if (int e = launch_split_weights(f2, f2_split, (long long)P * N * C, s)) return e;
        static const int ring_only = tune_env("MFTX_VOL_RING", 0);      // (tuning builds: the ring-buffered kernel, for A/B)
        if (tile_resident && !ring_only && volume_tile_applicable(C)) return launch_volume_tile(f1, f2_split, P, h, w, lvl, s);
        a.w = f2_split;
        a.arith = AR_SPLIT;
        return launch_cfg<128, 128, 4, 1, EPI_VOLUME, 32, AR_SPLIT, 2>(a, P, s, PC_CORR_VOLUME, 2.0 * N * N * (double)C * P);
    }
    return launch_cfg<128, 128, 4, 1, EPI_VOLUME>(a, P, s, PC_CORR_VOLUME, 2.0 * N * N * (double)C * P);
}

#endif  // MFTX_CONV_PART

}  // namespace mftx
