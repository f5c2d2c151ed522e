// Tile-resident convolution GEMM (split arithmetic): the layers of the update block whose input tile fits a CU's LDS.
//
// conv_gemm.hip streams BOTH operands through an LDS ring, chunk by chunk: per 32-wide K chunk every wave issues its share
// of the LDS-DMA pieces, waits for them, meets the others at a barrier -- the matrix pipes idle for a third of the loop
// (DESIGN.md section 4).  For a 3 x 3 / 1 x 5 / 5 x 1 convolution over <= 256 split-form channels the WHOLE input of an
// output tile of 128 cells -- halo included -- is 95-150 KB: it is loaded into LDS once, and then the K loop is nothing but
//     8 ds_read_b128 (A fragments at compile-time offsets) + 2 global loads (the wave's weight fragments, streamed from
//     L2 straight into registers, three steps ahead: no other wave of the workgroup reads them) + 12 MFMAs
// per 16-wide k group and wave, without a barrier, a DMA or a counted wait in it.  Measured on the same structure in
// flow_branch.hip: 0.82 of the matrix pipe inside the loop.
//
//   tile      TH x TW = 128 output cells: 8 x 16 (3 x 3), 4 x 32 (1 x 5), 32 x 4 (5 x 1); input halo tile in LDS, cell by
//             cell [C0 channels of segment 0 | C1 of segment 1 | 16 bytes], split form, zeros outside the image;
//   product   D = W x A^T (weights as the MFMA's first operand): a lane holds 4 x 4 consecutive output channels of one
//             cell per 32-cell row tile;
//   waves     N = 256: wave = one 32-channel column tile, all of K; N = 128: wave = (column tile, K half) -- channel
//             groups of the wave's parity -- the halves meet in LDS and are summed in a fixed order;
//   epilogue  every wave parks acc + accx / 2048 in LDS ([cell][channel] fp32, in the space of the input tile), then all
//             512 threads walk it row-wise, 8 consecutive channels of a cell each: bias / addend, activation or GRU gate
//             algebra (core/update.py:108-123), 32 contiguous bytes per lane out.
//
// Results do not depend on the batch or on a cell's place in its tile (one fixed sequence of products and sums per
// output).  They differ from conv_gemm.hip's in the order of the K sum (tap-major here as there, but the cross terms
// go to their accumulator in another order for N = 128): fp32 rounding.
#include "common.h"
#include "profile.h"

namespace mftx {

typedef float tc_f32x16 __attribute__((ext_vector_type(16)));
typedef float tc_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned tc_u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 tc_f16x8 __attribute__((ext_vector_type(8)));

enum TcEpi { TC_LINEAR = 0, TC_RELU = 1, TC_GRU_ZR = 2, TC_GRU_Q = 3,
             TC_RELU_PROJ = 4 };     // relu(. + bias) is not stored: it is multiplied, in place, with the NEXT layer's 3 x 3 x 2 filter (see below)

// Tuning builds only (-DMFTX_LF_TRACE): s_memtime stamps of workgroup 0's waves at the phase boundaries (tools/tc_trace.py)
#ifdef MFTX_LF_TRACE
__device__ unsigned long long tc_trace_buf[8][16];
#define TC_T(code) do { if (blockIdx.x == 0 && tcount < 16) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                        if ((threadIdx.x & 63) == 0) tc_trace_buf[threadIdx.x >> 6][tcount] = ((unsigned long long)(code) << 56) | (t_ & 0x00ffffffffffffffull); ++tcount; } } while (0)
// ... and, for EVERY workgroup, first / last stamps of both clocks: s_memtime counts shader cycles, s_memrealtime the constant
// 100 MHz reference -- their ratio is the shader clock the workgroup actually ran at (tools/clock_probe.py)
__device__ unsigned long long tc_clock_buf[4096][4];
#define TC_CLK(slot) do { if (wv == ((slot) ? 7 : 0) && blockIdx.x < 4096) { unsigned long long t_, r_; \
                          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(r_) :: "memory"); \
                          if (lane == 0) { tc_clock_buf[blockIdx.x][slot] = t_; tc_clock_buf[blockIdx.x][2 + slot] = r_; } } } while (0)
#else
#define TC_T(code) do { } while (0)
#define TC_CLK(slot) do { } while (0)
#endif

struct TileConvArgs {
    const float *a0; int lda0;      // segment 0: 128 channels per cell, split form, at a0 + cell * lda0 floats
    const float *a1; int lda1;      // segment 1 (128 more channels) or unused
    const void *wf;                 // mftx_pack_tile_conv_weights
    const float *bias;              // [N] or null
    const float *addend; int ld_addend;     // pre-activation addend [M][N] fp32 or null
    float *out; int ldo; int out_split;     // TC_LINEAR / TC_RELU
    float *z, *rh, *hf, *hx; int ld_hf, ld_hx;      // GRU epilogues (conv_gemm.hip: GruEpilogue)
    const void *wproj; float *tout;                 // TC_RELU_PROJ: the next layer's filter (launch_pack_flow_head) and the [M][18] partial products
    int P, h, w, tiles_x, tiles_y;
};

__device__ __forceinline__ void tc_barrier() {
    // s_waitcnt lgkmcnt(0): gfx950 has back-off barriers, so the compiler inserts NO wait in front of s_barrier and the builtin is no
    // fence -- without this a wave's last ds_write may still sit in the LDS queue when another wave reads the slot behind the
    // barrier (found in round 5 with tools/race_kernels.py: harmless with the GPU to itself, wrong values under contention).
    // LDS only: global prefetches and LDS-DMA loads (vmcnt) stay in flight, their consumers count them themselves.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ tc_f32x16 tc_mfma(const tc_f16x8 &a, const tc_f16x8 &b, const tc_f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// hi / lo halves of 8 consecutive values (conv_gemm.hip: split8)
__device__ __forceinline__ void tc_split8(const tc_f32x4 &u, const tc_f32x4 &v, float k2048, tc_u32x4 &hi, tc_u32x4 &lo) {
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
        "v_fma_mixhi_f16 %7, %15, %24, 0"
        : "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3),
          "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
        : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "s"(k2048));
    hi = tc_u32x4{h0, h1, h2, h3};
    lo = tc_u32x4{l0, l1, l2, l3};
}

// ... and of 4 consecutive values (the same operations per value: the same bits)
__device__ __forceinline__ void tc_split4(const tc_f32x4 &u, float k2048, unsigned (&hi)[2], unsigned (&lo)[2]) {
    unsigned h0, h1, l0, l1;
    float r0, r1, r2, r3;
    asm("v_cvt_pk_f16_f32 %0, %8, %9\n\t"
        "v_cvt_pk_f16_f32 %1, %10, %11\n\t"
        "v_fma_mix_f32 %4, %0, -1.0, %8 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %5, %0, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %6, %1, -1.0, %10 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %7, %1, -1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %2, %4, %12, 0\n\t"
        "v_fma_mixlo_f16 %3, %6, %12, 0\n\t"
        "v_fma_mixhi_f16 %2, %5, %12, 0\n\t"
        "v_fma_mixhi_f16 %3, %7, %12, 0"
        : "=&v"(h0), "=&v"(h1), "=&v"(l0), "=&v"(l1), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
        : "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "s"(k2048));
    hi[0] = h0; hi[1] = h1; lo[0] = l0; lo[1] = l1;
}

// the gate algebra, as conv_gemm.hip spells it (same functions: the two kernels agree to fp32 rounding of the K sum)
__device__ __forceinline__ float tc_sigmoid(float s) { return __frcp_rn(1.f + __expf(-s)); }
__device__ __forceinline__ float tc_tanh(float s) {
    const float t = __expf(-2.f * fabsf(s));
    return copysignf((1.f - t) * __frcp_rn(1.f + t), s);
}
__device__ __forceinline__ float tc_blend(float z, float h, float q) { return __fmaf_rn(z, q, __fmul_rn(__fsub_rn(1.f, z), h)); }

template <int TH, int TW, int KH, int KW, int CIN, int N>
struct TcGeom {
    static constexpr int CELLS = TH * TW, RT = CELLS / 32;  // output cells of a tile: 128, or 64 / 32 where 128-cell tiles would leave the chip empty; MFMA row tiles per wave
    static constexpr int HH = TH + KH - 1, HWD = TW + KW - 1, HCELLS = HH * HWD;
    static constexpr int CELLB = CIN * 4 + 16;              // consecutive cells start an odd number of 16-byte slots apart
    static constexpr int NT = N / 32, KS = 8 / NT;          // column tiles; K splits (waves per column tile)
    static constexpr int CG = CIN / 16, GPW = CG / KS;      // channel groups per tap; of them per wave
    static constexpr int STEPS = KH * KW * GPW;             // 16-wide k groups per wave
    static constexpr int RED_ROW = N + 4;                   // floats per cell of the parked sums
    static constexpr int A_BYTES = HCELLS * CELLB, RED_BYTES = KS * CELLS * RED_ROW * 4;
    static constexpr int LDS = A_BYTES > RED_BYTES ? A_BYTES : RED_BYTES;
    static constexpr int PROJ_ROW = 20, PROJ_BYTES = 2 * CELLS * PROJ_ROW * 4;        // TC_RELU_PROJ: two K halves of [cell][18 (+ 2)] behind the parked sums
    static_assert((CELLS == 128 || CELLS == 64 || CELLS == 32) && (N == 128 || N == 256) && (CIN == 128 || CIN == 256 || CIN == 144), "tile_conv: shapes");
    static_assert(LDS <= 160 * 1024, "tile_conv: the input tile must fit the CU's LDS");
};

template <int TH, int TW, int KH, int KW, int CIN, int N, int EPI>
__global__ __launch_bounds__(512, 2) void tile_conv_kernel(TileConvArgs p) {
    using G = TcGeom<TH, TW, KH, KW, CIN, N>;
    extern __shared__ __attribute__((aligned(16))) unsigned char tc_lds[];
    unsigned char *lds = tc_lds;
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = (int)blockIdx.x;
    const int tx_ = tile % p.tiles_x, ty_ = (tile / p.tiles_x) % p.tiles_y, img = tile / (p.tiles_x * p.tiles_y);
    const int x0 = tx_ * TW, y0 = ty_ * TH;
    const long long img_base = (long long)img * p.h * p.w;
    const int nt = wv % G::NT, ks = wv / G::NT;
    const uint4 *__restrict__ w2 = reinterpret_cast<const uint4 *>(p.wf) + (long long)((nt * G::KS + ks) * G::STEPS) * 128 + lane;

#ifdef MFTX_LF_TRACE
    int tcount = 0;
#endif
    TC_T(1);
    TC_CLK(0);
    // weight fragments of the first steps: in flight while the input tile loads
    constexpr int PF = 3;
    uint4 bq[PF][2];
#pragma unroll
    for (int s = 0; s < PF; ++s) { bq[s][0] = w2[s * 128]; bq[s][1] = w2[s * 128 + 64]; }

    // ---- the input tile (halo included) -> LDS, 16-byte pieces, zeros outside the image
    {
        constexpr int PPC = CIN / 4;                         // pieces per cell
        constexpr int TOTAL = G::HCELLS * PPC, ROUNDS = (TOTAL + 511) / 512, B = 6;
#pragma unroll 1
        for (int r0 = 0; r0 < ROUNDS; r0 += B) {
            uint4 v[B];
#pragma unroll
            for (int k = 0; k < B; ++k) {
                const int q = (r0 + k) * 512 + tid;
                v[k] = make_uint4(0u, 0u, 0u, 0u);
                if (r0 + k < ROUNDS && q < TOTAL) {
                    const int c = q / PPC, pc = q - c * PPC;
                    const int cy = c / G::HWD, cx = c - cy * G::HWD;
                    const int yy = y0 - KH / 2 + cy, xx = x0 - KW / 2 + cx;
                    if (yy >= 0 && yy < p.h && xx >= 0 && xx < p.w) {
                        const long long cell = img_base + (long long)yy * p.w + xx;
                        const float *src = (CIN == 256 && pc >= 32) ? p.a1 + cell * p.lda1 + (pc - 32) * 4 : p.a0 + cell * p.lda0 + pc * 4;
                        v[k] = *reinterpret_cast<const uint4 *>(src);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < B; ++k) {
                const int q = (r0 + k) * 512 + tid;
                if (r0 + k < ROUNDS && q < TOTAL) {
                    const int c = q / PPC, pc = q - c * PPC;
                    *reinterpret_cast<uint4 *>(lds + c * G::CELLB + pc * 16) = v[k];
                }
            }
        }
    }
    TC_T(2);
    tc_barrier();
    TC_T(3);

    // ---- the K loop
    tc_f32x16 acc[G::RT], accx[G::RT];
#pragma unroll
    for (int i = 0; i < G::RT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; accx[i][r] = 0.f; }
    const unsigned char *abase[G::RT];
    {
        const int r = lane & 31;
#pragma unroll
        for (int i = 0; i < G::RT; ++i) {
            const int m = 32 * i + r;
            abase[i] = lds + ((m / TW) * G::HWD + (m % TW)) * G::CELLB + (lane >> 5) * 32 + ks * 64;
        }
    }
    tc_f16x8 ah[2][G::RT], al[2][G::RT];
    auto read_a = [&](int s, int set) {
        const int tap = s / G::GPW, gg = s % G::GPW;
        const int off = ((tap / KW) * G::HWD + tap % KW) * G::CELLB + gg * 64 * G::KS;
#pragma unroll
        for (int i = 0; i < G::RT; ++i) {
            ah[set][i] = *reinterpret_cast<const tc_f16x8 *>(abase[i] + off);
            al[set][i] = *reinterpret_cast<const tc_f16x8 *>(abase[i] + off + 16);
        }
    };
    read_a(0, 0);
#pragma unroll
    for (int s = 0; s < G::STEPS; ++s) {
        const int set = s & 1;
        const tc_f16x8 bh = __builtin_bit_cast(tc_f16x8, bq[s % PF][0]), bl = __builtin_bit_cast(tc_f16x8, bq[s % PF][1]);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < G::STEPS) read_a(s + 1, set ^ 1);
        if (s + PF < G::STEPS) { bq[s % PF][0] = w2[(s + PF) * 128]; bq[s % PF][1] = w2[(s + PF) * 128 + 64]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < G::RT; ++i) acc[i] = tc_mfma(bh, ah[set][i], acc[i]);
#pragma unroll
        for (int i = 0; i < G::RT; ++i) accx[i] = tc_mfma(bl, ah[set][i], accx[i]);
#pragma unroll
        for (int i = 0; i < G::RT; ++i) accx[i] = tc_mfma(bh, al[set][i], accx[i]);
        __builtin_amdgcn_sched_barrier(0);
    }
    TC_T(4);
    tc_barrier();           // every wave is done with the input tile: its space takes the sums
    TC_T(5);

    // ---- sums -> LDS [K split][cell][channel]
    const float inv2048 = 1.f / 2048.f;
    float *red = reinterpret_cast<float *>(lds);
#pragma unroll
    for (int i = 0; i < G::RT; ++i)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            tc_f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][4 * b + e] + accx[i][4 * b + e] * inv2048;
            *reinterpret_cast<tc_f32x4 *>(red + (ks * G::CELLS + 32 * i + (lane & 31)) * G::RED_ROW + 32 * nt + 8 * b + 4 * (lane >> 5)) = v;
        }
    TC_T(6);
    tc_barrier();
    TC_T(7);

    const float k2048 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(0x45000000));
    if constexpr (EPI == TC_RELU_PROJ) {
        // The flow head (core/update.py:6-14): delta = conv2(relu(conv1(h))), conv2 a 3 x 3 filter with TWO outputs.  Its 256
        // input channels are this kernel's output, and by linearity
        //     delta[c][o] = b2[o] + sum over taps t of T[c + offset(t)][t][o],   T[c'][t][o] = sum_k relu(conv1)[c'][k] W2[o][k][t]:
        // the 18 numbers T[c'][.][.] need cell c' alone.  So relu(conv1) -- 29 MB per iteration at 7 pairs of 512 x 512 -- is
        // never stored: the tile's [128 cells x 256] block, parked in LDS, is multiplied here with W2 as a [256 x 18] matrix
        // (split arithmetic; 192 more MFMAs after the layer's 6912) and only T leaves (2 MB); flow_head_sum_kernel adds the
        // nine shifted terms.  Wave (row tile mt, K half kh): 8 k groups; the halves meet in LDS, summed in a fixed order.
        const int mt = wv & 3, kh = wv >> 2;         // (tiles of fewer than 128 cells: the waves of the missing row tiles only keep the barriers)
        const uint4 *__restrict__ wp = reinterpret_cast<const uint4 *>(p.wproj) + lane;
        float *tp = reinterpret_cast<float *>(lds + G::RED_BYTES);
        if (mt < G::RT) {
        tc_f32x16 d, dx;
#pragma unroll
        for (int r = 0; r < 16; ++r) { d[r] = 0.f; dx[r] = 0.f; }
        const float *xrow = red + (32 * mt + (lane & 31)) * G::RED_ROW + 8 * (lane >> 5);
#pragma unroll
        for (int gg = 0; gg < 8; ++gg) {
            const int g = 8 * kh + gg;
            tc_f32x4 u = *reinterpret_cast<const tc_f32x4 *>(xrow + 16 * g), v = *reinterpret_cast<const tc_f32x4 *>(xrow + 16 * g + 4);
            u += *reinterpret_cast<const tc_f32x4 *>(p.bias + 16 * g + 8 * (lane >> 5));
            v += *reinterpret_cast<const tc_f32x4 *>(p.bias + 16 * g + 8 * (lane >> 5) + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { u[e] = relu_keep_nan(u[e]); v[e] = relu_keep_nan(v[e]); }
            tc_u32x4 hi, lo;
            tc_split8(u, v, k2048, hi, lo);
            const tc_f16x8 wh = __builtin_bit_cast(tc_f16x8, wp[(g * 2) * 64]), wl = __builtin_bit_cast(tc_f16x8, wp[(g * 2 + 1) * 64]);
            asm volatile("s_nop 1" : "+v"(hi), "+v"(lo));      // (the split's results, two wait states before the MFMA reads them)
            const tc_f16x8 xh = __builtin_bit_cast(tc_f16x8, hi), xl = __builtin_bit_cast(tc_f16x8, lo);
            d = tc_mfma(wh, xh, d);
            dx = tc_mfma(wl, xh, dx);
            dx = tc_mfma(wh, xl, dx);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j0 = 8 * b + 4 * (lane >> 5);          // this lane's outputs j0 .. j0 + 3 of cell 32 mt + (lane & 31); 18 exist
            if (j0 < G::PROJ_ROW) {
                tc_f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = d[4 * b + e] + dx[4 * b + e] * inv2048;
                *reinterpret_cast<tc_f32x4 *>(tp + (kh * G::CELLS + 32 * mt + (lane & 31)) * G::PROJ_ROW + j0) = v;
            }
        }
        }
        tc_barrier();
        for (int idx = tid; idx < G::CELLS * 18; idx += 512) {
            const int m = idx / 18, j = idx - m * 18;
            const int yy = y0 + m / TW, xx = x0 + m % TW;
            if (yy < p.h && xx < p.w)
                p.tout[(img_base + (long long)yy * p.w + xx) * 18 + j] = tp[m * G::PROJ_ROW + j] + tp[(G::CELLS + m) * G::PROJ_ROW + j];
        }
        TC_T(8);
        TC_CLK(1);
        return;
    }

    // ---- row-wise epilogue: 8 consecutive channels of a cell per lane
    constexpr int GPC = N / 8, ITEMS = G::CELLS * GPC;
#pragma unroll
    for (int it = 0; it < ITEMS / 512; ++it) {
        const int item = tid + 512 * it, m = item / GPC, n0 = (item % GPC) * 8;
        const int yy = y0 + m / TW, xx = x0 + m % TW;
        const float *src = red + m * G::RED_ROW + n0;
        tc_f32x4 u = *reinterpret_cast<const tc_f32x4 *>(src), v = *reinterpret_cast<const tc_f32x4 *>(src + 4);
        if constexpr (G::KS == 2) {
            u += *reinterpret_cast<const tc_f32x4 *>(src + G::CELLS * G::RED_ROW);
            v += *reinterpret_cast<const tc_f32x4 *>(src + G::CELLS * G::RED_ROW + 4);
        }
        if (yy >= p.h || xx >= p.w) continue;
        const long long cell = img_base + (long long)yy * p.w + xx;
        if (p.bias) {
            u += *reinterpret_cast<const tc_f32x4 *>(p.bias + n0);
            v += *reinterpret_cast<const tc_f32x4 *>(p.bias + n0 + 4);
        }
        if (p.addend) {
            u += *reinterpret_cast<const tc_f32x4 *>(p.addend + cell * p.ld_addend + n0);
            v += *reinterpret_cast<const tc_f32x4 *>(p.addend + cell * p.ld_addend + n0 + 4);
        }
        auto store_split = [&](float *row, int c0) {        // 8 channels c0 .. c0 + 7 (c0 % 8 == 0) of a split-form row
            tc_u32x4 hi, lo;
            tc_split8(u, v, k2048, hi, lo);
            uint4 *dst = reinterpret_cast<uint4 *>(reinterpret_cast<char *>(row) + (c0 >> 3) * 32);
            dst[0] = __builtin_bit_cast(uint4, hi);
            dst[1] = __builtin_bit_cast(uint4, lo);
        };
        if constexpr (EPI == TC_GRU_ZR) {                   // [z | r] gates; r is folded into r * h (core/update.py:113-115, 119-121)
#pragma unroll
            for (int e = 0; e < 4; ++e) { u[e] = tc_sigmoid(u[e]); v[e] = tc_sigmoid(v[e]); }
            if (n0 < 128) {
                *reinterpret_cast<tc_f32x4 *>(p.z + cell * 128 + n0) = u;
                *reinterpret_cast<tc_f32x4 *>(p.z + cell * 128 + n0 + 4) = v;
            } else {
                u *= *reinterpret_cast<const tc_f32x4 *>(p.hf + cell * p.ld_hf + n0 - 128);
                v *= *reinterpret_cast<const tc_f32x4 *>(p.hf + cell * p.ld_hf + n0 - 128 + 4);
                store_split(p.rh + cell * 128, n0 - 128);
            }
        } else if constexpr (EPI == TC_GRU_Q) {             // candidate q, h <- (1 - z) h + z q (core/update.py:116-117, 122-123)
            const tc_f32x4 z0 = *reinterpret_cast<const tc_f32x4 *>(p.z + cell * 128 + n0), z1 = *reinterpret_cast<const tc_f32x4 *>(p.z + cell * 128 + n0 + 4);
            float *hrow = p.hf + cell * p.ld_hf + n0;
            const tc_f32x4 h0 = *reinterpret_cast<const tc_f32x4 *>(hrow), h1 = *reinterpret_cast<const tc_f32x4 *>(hrow + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { u[e] = tc_blend(z0[e], h0[e], tc_tanh(u[e])); v[e] = tc_blend(z1[e], h1[e], tc_tanh(v[e])); }
            *reinterpret_cast<tc_f32x4 *>(hrow) = u;
            *reinterpret_cast<tc_f32x4 *>(hrow + 4) = v;
            store_split(p.hx + cell * p.ld_hx, n0);
        } else {
            if constexpr (EPI == TC_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { u[e] = relu_keep_nan(u[e]); v[e] = relu_keep_nan(v[e]); }
            }
            if (p.out_split) store_split(p.out + cell * p.ldo, n0);
            else {
                *reinterpret_cast<tc_f32x4 *>(p.out + cell * p.ldo + n0) = u;
                *reinterpret_cast<tc_f32x4 *>(p.out + cell * p.ldo + n0 + 4) = v;
            }
        }
    }
    TC_T(8);
    TC_CLK(1);
}

#ifdef MFTX_LF_TRACE
extern "C" int mftx_debug_tc_clock(unsigned long long *out, int n_wg) {
    if (n_wg < 0 || n_wg > 4096) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(tc_clock_buf), sizeof(unsigned long long) * 4 * n_wg) == hipSuccess ? 0 : -1;
}
extern "C" int mftx_debug_tc_trace(unsigned long long *out) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(tc_trace_buf), sizeof(unsigned long long) * 8 * 16) != hipSuccess) return -1;
    unsigned long long z[8 * 16] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(tc_trace_buf), z, sizeof z) == hipSuccess ? 0 : -1;
}
#endif

// ---- one GRU half in ONE kernel (core/update.py:108-123): z | r gates -> r * h -> candidate q -> h <- (1 - z) h + z q -------------
// The two launches of a SepConvGRU pass share their input: [h | motion] of a tile feeds the z | r gates, [r * h | motion] of the
// same tile the candidate.  Here the tile is loaded ONCE; the gates' K loop runs as in tile_conv_kernel<..., 256, 256>, r * h
// replaces h IN PLACE in the LDS tile (the fp32 sums of 8 channels and their split form are the same 32 bytes), the
// candidate's K loop runs as in tile_conv_kernel<..., 256, 128> over the same tile, and the blend writes the new h: one
// launch, one load phase, and r * h never leaves the CU.
//   halo      the candidate at a cell needs r * h at its +- 2 neighbours along the pass: a tile's 128 cells are the "R cells"
//             where the gates are evaluated; cells within 2 of an R edge that is not an image border are evaluated for their
//             neighbours' sake only, their own new h comes from the adjacent tile (tiles advance by TW - 4 along the pass; a
//             tile that spans the whole image width -- 64 cells at 512 x 512 -- recomputes nothing);
//   h         is read from one buffer and written to another (h_in / h_out, hf_in / hf_out): a workgroup's halo cells are its neighbours'
//             outputs, and with more tiles than CUs a late workgroup would find its halo already updated;
//   z         goes through global memory (64 KB per tile, written and read by the same workgroup: L2) -- LDS is full;
//   sums      every output is the same sequence of products and sums as in the two kernels this replaces: same bits.
struct GruHalfArgs {
    const float *h_in; int ld_hin;      // h, split form, 128 channels at h_in + cell * ld_hin floats
    const float *mo; int ld_mo;         // motion features, split form, 128 channels
    const void *wzr, *wq;               // mftx_pack_tile_conv_weights streams: [z | r] (N = 256) and q (N = 128), cin = 256
    const float *pre_zr, *pre_q;        // the gates' context parts + bias (pre-activation addends): [M][256], [M][128]
    float *z;                           // scratch [M][128]
    const float *hf_in; float *hf_out;  // h in fp32 [M][128]: read from one buffer, written to another, like the split form
    float *h_out; int ld_hout;          // new h, split form
    int P, h, w, tiles_x, tiles_y, step;
};

template <class G, int KW>
__device__ __forceinline__ void tc_kloop(const unsigned char *const (&abase)[G::RT], const uint4 *__restrict__ w2, uint4 (&bq)[3][2],
                                         tc_f32x16 (&acc)[G::RT], tc_f32x16 (&accx)[G::RT]) {
    constexpr int PF = 3;
    tc_f16x8 ah[2][G::RT], al[2][G::RT];
    auto read_a = [&](int s, int set) {
        const int tap = s / G::GPW, gg = s % G::GPW;
        const int off = ((tap / KW) * G::HWD + tap % KW) * G::CELLB + gg * 64 * G::KS;
#pragma unroll
        for (int i = 0; i < G::RT; ++i) {
            ah[set][i] = *reinterpret_cast<const tc_f16x8 *>(abase[i] + off);
            al[set][i] = *reinterpret_cast<const tc_f16x8 *>(abase[i] + off + 16);
        }
    };
    read_a(0, 0);
#pragma unroll
    for (int s = 0; s < G::STEPS; ++s) {
        const int set = s & 1;
        const tc_f16x8 bh = __builtin_bit_cast(tc_f16x8, bq[s % PF][0]), bl = __builtin_bit_cast(tc_f16x8, bq[s % PF][1]);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < G::STEPS) read_a(s + 1, set ^ 1);
        if (s + PF < G::STEPS) { bq[s % PF][0] = w2[(s + PF) * 128]; bq[s % PF][1] = w2[(s + PF) * 128 + 64]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < G::RT; ++i) acc[i] = tc_mfma(bh, ah[set][i], acc[i]);
#pragma unroll
        for (int i = 0; i < G::RT; ++i) accx[i] = tc_mfma(bl, ah[set][i], accx[i]);
#pragma unroll
        for (int i = 0; i < G::RT; ++i) accx[i] = tc_mfma(bh, al[set][i], accx[i]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int TH, int TW, int KH, int KW>
__global__ __launch_bounds__(512, 2) void gru_half_kernel(GruHalfArgs p) {
    using G1 = TcGeom<TH, TW, KH, KW, 256, 256>;            // the gates: wave = one 32-channel column tile of [z | r], all of K
    using G2 = TcGeom<TH, TW, KH, KW, 256, 128>;            // the candidate: wave = (column tile, K half)
    static_assert(G1::CELLB == G2::CELLB && G1::HWD == G2::HWD, "one tile, two GEMMs");
    constexpr int CELLB = G1::CELLB, HWD = G1::HWD, PF = 3, RT = G1::RT, CELLS = G1::CELLS;
    constexpr bool HORIZ = KW > 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char tc_lds[];
    unsigned char *lds = tc_lds;
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = (int)blockIdx.x;
    const int tx_ = tile % p.tiles_x, ty_ = (tile / p.tiles_x) % p.tiles_y, img = tile / (p.tiles_x * p.tiles_y);
    // the R cells' origin, and the cells of them whose new h this workgroup writes: [olo, ohi) along the pass
    const int n_along = HORIZ ? p.tiles_x : p.tiles_y, t_along = HORIZ ? tx_ : ty_, len_along = HORIZ ? p.w : p.h;
    const int r0 = n_along > 1 ? t_along * p.step - 2 : 0;
    const int olo = n_along > 1 ? t_along * p.step : 0, ohi = n_along > 1 ? min(len_along, olo + p.step) : len_along;
    const int x0 = HORIZ ? r0 : tx_ * TW, y0 = HORIZ ? ty_ * TH : r0;
    const long long img_base = (long long)img * p.h * p.w;
    const uint4 *__restrict__ w1 = reinterpret_cast<const uint4 *>(p.wzr) + (long long)((wv ^ 4) * G1::STEPS) * 128 + lane;   // (r for waves 0-3: see below)
    const int nt2 = wv % G2::NT, ks2 = wv / G2::NT;
    const uint4 *__restrict__ w2 = reinterpret_cast<const uint4 *>(p.wq) + (long long)((nt2 * G2::KS + ks2) * G2::STEPS) * 128 + lane;

#ifdef MFTX_LF_TRACE
    int tcount = 0;
#endif
    TC_T(1);
    TC_CLK(0);
    uint4 bq[PF][2];
#pragma unroll
    for (int s = 0; s < PF; ++s) { bq[s][0] = w1[s * 128]; bq[s][1] = w1[s * 128 + 64]; }

    // ---- [h | motion] of the tile (halo included) -> LDS, 16-byte pieces, zeros outside the image
    {
        constexpr int PPC = 64, TOTAL = G1::HCELLS * PPC, ROUNDS = (TOTAL + 511) / 512, B = 6;
#pragma unroll 1
        for (int rr = 0; rr < ROUNDS; rr += B) {
            uint4 v[B];
#pragma unroll
            for (int k = 0; k < B; ++k) {
                const int q = (rr + k) * 512 + tid;
                v[k] = make_uint4(0u, 0u, 0u, 0u);
                if (rr + k < ROUNDS && q < TOTAL) {
                    const int c = q / PPC, pc = q - c * PPC;
                    const int cy = c / HWD, cx = c - cy * HWD;
                    const int yy = y0 - KH / 2 + cy, xx = x0 - KW / 2 + cx;
                    if (yy >= 0 && yy < p.h && xx >= 0 && xx < p.w) {
                        const long long cell = img_base + (long long)yy * p.w + xx;
                        const float *src = pc >= 32 ? p.mo + cell * p.ld_mo + (pc - 32) * 4 : p.h_in + cell * p.ld_hin + pc * 4;
                        v[k] = *reinterpret_cast<const uint4 *>(src);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < B; ++k) {
                const int q = (rr + k) * 512 + tid;
                if (rr + k < ROUNDS && q < TOTAL) {
                    const int c = q / PPC, pc = q - c * PPC;
                    *reinterpret_cast<uint4 *>(lds + c * CELLB + pc * 16) = v[k];
                }
            }
        }
    }
    TC_T(2);
    tc_barrier();
    TC_T(3);

    tc_f32x16 acc[RT], accx[RT];
    const unsigned char *abase[RT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; accx[i][r] = 0.f; }
    };
    auto set_abase = [&](int ks) {
        const int r = lane & 31;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int m = 32 * i + r;
            abase[i] = lds + ((m / TW) * HWD + (m % TW)) * CELLB + (lane >> 5) * 32 + ks * 64;
        }
    };
    // an R cell's own slot in the tile: its h part is the first 512 bytes, [8 channels: hi x 8 | lo x 8] x 16
    auto slot = [&](int m) { return lds + ((m / TW + KH / 2) * HWD + (m % TW + KW / 2)) * CELLB; };
    const float inv2048 = 1.f / 2048.f;
    const float k2048 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(0x45000000));

    // ---- z | r gates.  The r waves are the OLDER wave of every SIMD (column tile wv ^ 4: waves 0-3 take r, waves 4-7 take z): the
    // matrix pipe serves the older wave first, so it leaves this K loop ~27 k cycles before its partner (tools/gru_trace.py) -- and r is
    // the gate with work on the critical path behind it.
    const int nt1 = wv ^ 4;
    zero_acc();
    set_abase(0);
    tc_kloop<G1, KW>(abase, w1, bq, acc, accx);
    TC_T(4);
    // (the candidate's first weight fragments: in flight during the gate algebra)
#pragma unroll
    for (int s = 0; s < PF; ++s) { bq[s][0] = w2[s * 128]; bq[s][1] = w2[s * 128 + 64]; }
    // r * h by the r waves themselves, straight from their accumulators (a lane: 4 x 4 consecutive channels of one cell per row tile),
    // in the shadow of their SIMD partners' K loop: sigmoid(. + context part) * h, split, KEPT IN REGISTERS (the accumulators are dead) --
    // the h parts of the tile are still being read.  Same arithmetic per value as the row-wise pass over parked sums this replaces
    // (8.5 k exposed cycles of the launch's 144 k between two barriers): the same bits.
    unsigned rh_hi[RT][4][2], rh_lo[RT][4][2];
    if (wv < 4) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int m = 32 * i + (lane & 31);
            const int yy = y0 + m / TW, xx = x0 + m % TW;
            const bool inside = yy >= 0 && yy < p.h && xx >= 0 && xx < p.w;
            const int yc = yy < 0 ? 0 : (yy >= p.h ? p.h - 1 : yy), xc = xx < 0 ? 0 : (xx >= p.w ? p.w - 1 : xx);
            const long long cell = img_base + (long long)yc * p.w + xc;
            const int n0 = 32 * wv + 4 * (lane >> 5);
            tc_f32x4 a[4], hh[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                a[b] = *reinterpret_cast<const tc_f32x4 *>(p.pre_zr + cell * 256 + 128 + n0 + 8 * b);
                hh[b] = *reinterpret_cast<const tc_f32x4 *>(p.hf_in + cell * 128 + n0 + 8 * b);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                tc_f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][4 * b + e] + accx[i][4 * b + e] * inv2048;
                v += a[b];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = tc_sigmoid(v[e]);
                v *= hh[b];
                tc_split4(v, k2048, rh_hi[i][b], rh_lo[i][b]);
                if (!inside) { rh_hi[i][b][0] = rh_hi[i][b][1] = rh_lo[i][b][0] = rh_lo[i][b][1] = 0u; }     // (the candidate's zero padding)
            }
        }
    }
    TC_T(5);
    tc_barrier();           // every wave is done with the h parts of the tile
    // r * h, in place: a cell's 8-channel group is [hi x 8 | lo x 8]; this lane holds channels 4 (lane >> 5) .. + 3 of group 4 wv + b
    if (wv < 4) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            unsigned char *dst = slot(32 * i + (lane & 31)) + (4 * wv) * 32 + (lane >> 5) * 8;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                *reinterpret_cast<uint2 *>(dst + 32 * b) = make_uint2(rh_hi[i][b][0], rh_hi[i][b][1]);
                *reinterpret_cast<uint2 *>(dst + 32 * b + 16) = make_uint2(rh_lo[i][b][0], rh_lo[i][b][1]);
            }
        }
    }
    tc_barrier();
    TC_T(6);
    // The z waves' sums (acc + accx / 2048, before the context part and the sigmoid) -> global, straight from the accumulators, while
    // their SIMD partners -- older, served first anyway -- are already in the candidate's K loop; stores only: the context part and the
    // sigmoid wait for the blend, whose row-wise pass reads z back anyway (same arithmetic per value: the same bits).
    if (wv >= 4) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int m = 32 * i + (lane & 31);
            const int yy = y0 + m / TW, xx = x0 + m % TW, along = HORIZ ? xx : yy;
            const bool own = yy >= 0 && yy < p.h && xx >= 0 && xx < p.w && along >= olo && along < ohi;
            const long long cell = img_base + (long long)yy * p.w + xx;
            const int n0 = 32 * nt1 + 4 * (lane >> 5);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                tc_f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][4 * b + e] + accx[i][4 * b + e] * inv2048;
                if (own) *reinterpret_cast<tc_f32x4 *>(p.z + cell * 128 + n0 + 8 * b) = v;
            }
        }
    }

    // ---- the candidate over [r * h | motion]
    zero_acc();
    set_abase(ks2);
    tc_kloop<G2, KW>(abase, w2, bq, acc, accx);
    TC_T(7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (this wave's z stores have reached L2 before anyone reads z back)
    tc_barrier();           // every wave is done with the tile: its space takes the sums
    float *red = reinterpret_cast<float *>(lds);
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            tc_f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][4 * b + e] + accx[i][4 * b + e] * inv2048;
            *reinterpret_cast<tc_f32x4 *>(red + (ks2 * CELLS + 32 * i + (lane & 31)) * G2::RED_ROW + 32 * nt2 + 8 * b + 4 * (lane >> 5)) = v;
        }
    tc_barrier();
    // q = tanh(. + context part), h <- (1 - z) h + z q (core/update.py:116-117, 122-123): the cells this workgroup owns
#pragma unroll
    for (int it = 0; it < RT; ++it) {
        const int item = tid + 512 * it, m = item >> 4, n0 = (item & 15) * 8;
        const int yy = y0 + m / TW, xx = x0 + m % TW, along = HORIZ ? xx : yy;
        const float *src = red + m * G2::RED_ROW + n0;
        tc_f32x4 u = *reinterpret_cast<const tc_f32x4 *>(src), v = *reinterpret_cast<const tc_f32x4 *>(src + 4);
        u += *reinterpret_cast<const tc_f32x4 *>(src + CELLS * G2::RED_ROW);
        v += *reinterpret_cast<const tc_f32x4 *>(src + CELLS * G2::RED_ROW + 4);
        if (yy < 0 || yy >= p.h || xx < 0 || xx >= p.w || along < olo || along >= ohi) continue;
        const long long cell = img_base + (long long)yy * p.w + xx;
        u += *reinterpret_cast<const tc_f32x4 *>(p.pre_q + cell * 128 + n0);
        v += *reinterpret_cast<const tc_f32x4 *>(p.pre_q + cell * 128 + n0 + 4);
        tc_f32x4 z0 = *reinterpret_cast<const tc_f32x4 *>(p.z + cell * 128 + n0), z1 = *reinterpret_cast<const tc_f32x4 *>(p.z + cell * 128 + n0 + 4);
        z0 += *reinterpret_cast<const tc_f32x4 *>(p.pre_zr + cell * 256 + n0);
        z1 += *reinterpret_cast<const tc_f32x4 *>(p.pre_zr + cell * 256 + n0 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { z0[e] = tc_sigmoid(z0[e]); z1[e] = tc_sigmoid(z1[e]); }
        const float *hrow = p.hf_in + cell * 128 + n0;
        const tc_f32x4 h0 = *reinterpret_cast<const tc_f32x4 *>(hrow), h1 = *reinterpret_cast<const tc_f32x4 *>(hrow + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { u[e] = tc_blend(z0[e], h0[e], tc_tanh(u[e])); v[e] = tc_blend(z1[e], h1[e], tc_tanh(v[e])); }
        *reinterpret_cast<tc_f32x4 *>(p.hf_out + cell * 128 + n0) = u;
        *reinterpret_cast<tc_f32x4 *>(p.hf_out + cell * 128 + n0 + 4) = v;
        tc_u32x4 hi, lo;
        tc_split8(u, v, k2048, hi, lo);
        uint4 *dst = reinterpret_cast<uint4 *>(reinterpret_cast<char *>(p.h_out + cell * p.ld_hout) + (n0 >> 3) * 32);
        dst[0] = __builtin_bit_cast(uint4, hi);
        dst[1] = __builtin_bit_cast(uint4, lo);
    }
    TC_T(8);
    TC_CLK(1);
}

template <int TH, int TW, int KH, int KW>
static int gru_half_launch(GruHalfArgs a, hipStream_t s) {
    using G1 = TcGeom<TH, TW, KH, KW, 256, 256>;
    using G2 = TcGeom<TH, TW, KH, KW, 256, 128>;
    constexpr int lds_bytes = G1::A_BYTES > G2::RED_BYTES ? G1::A_BYTES : G2::RED_BYTES;      // the tile, then the candidate's parked sums
    static_assert(lds_bytes <= 160 * 1024, "gru_half: LDS");
    constexpr bool HORIZ = KW > 1;
    const int along = HORIZ ? a.w : a.h, T = HORIZ ? TW : TH;
    a.step = T - 4;
    const int n_along = along <= T ? 1 : cdiv(along, a.step);
    a.tiles_x = HORIZ ? n_along : cdiv(a.w, TW);
    a.tiles_y = HORIZ ? cdiv(a.h, TH) : n_along;
    const long long tiles = (long long)a.P * a.tiles_x * a.tiles_y;
    if (tiles > 0x7fffffffLL) return fail(MFTX_E_ARG, "gru_half: too many tiles");
    auto kern = gru_half_kernel<TH, TW, KH, KW>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
            return fail(MFTX_E_STATE, "gru_half: cannot reserve %d bytes of LDS", lds_bytes);
        attr_set = true;
    }
    ProfScope prof(PC_GRU_FUSED, s, 2.0 * a.P * a.h * a.w * 384.0 * KH * KW * 256);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), lds_bytes, s, a);
    return check_launch("gru_half");
}

static int tc_num_cus();
static bool tc_fills(long long tiles);
int tile_conv_cells(int P, int h, int w, int kh);
// tiles of the fused kernel for a pass over P maps of h x w cells with R tiles of th x tw cells
long long gru_half_tiles(int P, int h, int w, int th, int tw, bool horiz) {
    const int along = horiz ? w : h, T = horiz ? tw : th;
    const int n_along = along <= T ? 1 : cdiv(along, T - 4);
    return (long long)P * (horiz ? n_along * cdiv(h, th) : n_along * cdiv(w, tw));
}

int launch_gru_half(const GruHalfLaunch &d, hipStream_t s) {
    if (!d.h_in || !d.mo || !d.wzr || !d.wq || !d.pre_zr || !d.pre_q || !d.z || !d.hf_in || !d.hf_out || !d.h_out) return fail(MFTX_E_ARG, "gru_half: null pointer");
    if (d.P <= 0 || d.h <= 0 || d.w <= 0 || (d.pass != 0 && d.pass != 1)) return fail(MFTX_E_ARG, "gru_half: bad sizes");
    if (d.h_in == d.h_out || d.hf_in == d.hf_out) return fail(MFTX_E_ARG, "gru_half: h is read from one buffer and written to another");
    auto bad_split = [](const float *p, int ld) { return (reinterpret_cast<uintptr_t>(p) & 31) != 0 || (ld % 8) != 0; };
    if (bad_split(d.h_in, d.ld_hin) || bad_split(d.mo, d.ld_mo) || bad_split(d.h_out, d.ld_hout) || !aligned16(d.wzr) || !aligned16(d.wq) ||
        !aligned16(d.pre_zr) || !aligned16(d.pre_q) || !aligned16(d.z) || !aligned16(d.hf_in) || !aligned16(d.hf_out))
        return fail(MFTX_E_ALIGN, "gru_half: split-form rows are 32-byte aligned with strides in multiples of 8; the rest 16-byte aligned");
    GruHalfArgs a{};
    a.h_in = d.h_in; a.ld_hin = d.ld_hin; a.mo = d.mo; a.ld_mo = d.ld_mo; a.wzr = d.wzr; a.wq = d.wq; a.pre_zr = d.pre_zr; a.pre_q = d.pre_q;
    a.z = d.z; a.hf_in = d.hf_in; a.hf_out = d.hf_out; a.h_out = d.h_out; a.ld_hout = d.ld_hout; a.P = d.P; a.h = d.h; a.w = d.w;
    // R tiles of 128 cells where they fill the chip -- 2 x 64 (a 64-wide map is one tile per row pair: nothing is recomputed) or
    // 4 x 32, whichever needs fewer workgroups --, else of 64 or 32 cells: the same kernel with fewer row tiles per wave, the same bits
    const bool hz = d.pass == 0;
    auto best = [&](int cells, int &th, int &tw) {
        const int ta = hz ? cells / 64 : 64, tb = hz ? 64 : cells / 64;          // long tiles: 2 x 64, 1 x 64 | 64 x 2, 64 x 1
        const int sa = hz ? cells / 32 : 32, sb = hz ? 32 : cells / 32;          // short tiles: 4 x 32, 2 x 32, 1 x 32 | 32 x 4, 32 x 2, 32 x 1
        const long long nl = cells >= 64 ? gru_half_tiles(d.P, d.h, d.w, ta, tb, hz) : (1ll << 62), ns = gru_half_tiles(d.P, d.h, d.w, sa, sb, hz);
        if (nl <= ns) { th = ta; tw = tb; return nl; }
        th = sa; tw = sb; return ns;
    };
    int th = 0, tw = 0, cells = d.cells;
    if (!cells) {
        cells = 128;
        if (!tc_fills(best(128, th, tw))) cells = 64;
    }
    if (cells != 128 && cells != 64 && cells != 32) return fail(MFTX_E_ARG, "gru_half: 128, 64 or 32 cells per tile");
    if (cells == 32) cells = 64;         // (the 32-cell instances of this kernel spill -- the compiler's doing, not the algorithm's; 64 is the smallest built)
    best(cells, th, tw);
#define GH(TH, TW, KH, KW) if (th == TH && tw == TW) return gru_half_launch<TH, TW, KH, KW>(a, s)
    if (hz) { GH(2, 64, 1, 5); GH(4, 32, 1, 5); GH(1, 64, 1, 5); GH(2, 32, 1, 5); }
    else { GH(64, 2, 5, 1); GH(32, 4, 5, 1); GH(64, 1, 5, 1); GH(32, 2, 5, 1); }
#undef GH
    return fail(MFTX_E_STATE, "gru_half: no kernel for tiles of %d x %d cells", th, tw);
}

// ---- the occlusion + uncertainty heads in ONE kernel (core/update.py:177-214: two heads of conv3x3 712 -> 128, relu, conv3x3 -> 2 | 1) ----
// Their first layers share the 712-channel input and are one 712 -> 256 GEMM; on the ring-buffered kernel that GEMM re-read its
// 82 MB input nine times through L2 (360-720 MB fetched per launch) and a VALU kernel then read the 29 MB of hidden channels
// back for the 3 x 3 x 3 second layers.  Here: 712 channels do not fit LDS at once (180 halo cells x 2.8 KB), so the tile is
// loaded in FIVE channel passes of 144 (720 = 712 + 8 zero channels; 104 KB each), the accumulators live across the passes,
// and the second layers are a projection epilogue like the flow head's (TC_RELU_PROJ): relu(. + bias), parked in LDS, times the
// [256 x 27] matrix W2'[k][3 tap + o], 27 numbers per cell out -- ou_heads_sum_kernel adds the nine shifted terms.
struct OuHeadArgs {
    const float *a; int lda;        // the heads' input, split form, 712 channels at a + cell * lda floats -- or null: gathered from its parts
    // gather mode (the engine): [net128 | inp128] = hx[0:256] and motion128 = hx[256:384] (split form), corr324 (fp32), flow = coords1 - grid,
    // delta (core/update.py:197, core/raft.py:199-206) -- the concatenation is never materialised; flow_lr (the upsampler's input) is
    // written for the tile's own cells on the way
    const float *hx; const float *corr; int ld_corr; const float *coords1; const float *delta; float *flow_lr;
    const void *wf;                 // launch_pack_ou_head: [nt][pass][tap][group][hi | lo][lane] x 16 bytes
    const float *bias;              // [256]
    const void *wproj; float *tout; // launch_pack_proj27; [M][27]
    int P, h, w, tiles_x, tiles_y;
};
constexpr int OU_C = 712, OU_PASSES = 5, OU_CP = 144, OU_PROJ_ROW = 28;

template <int TH, int TW, bool GATHER>
__global__ __launch_bounds__(512, 2) void ou_head_kernel(OuHeadArgs p) {
    using G = TcGeom<TH, TW, 3, 3, OU_CP, 256>;
    constexpr int RT = G::RT, CELLS = G::CELLS, PF = 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char tc_lds[];
    unsigned char *lds = tc_lds;
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = (int)blockIdx.x;
    const int tx_ = tile % p.tiles_x, ty_ = (tile / p.tiles_x) % p.tiles_y, img = tile / (p.tiles_x * p.tiles_y);
    const int x0 = tx_ * TW, y0 = ty_ * TH;
    const long long img_base = (long long)img * p.h * p.w;
    const uint4 *__restrict__ w2 = reinterpret_cast<const uint4 *>(p.wf) + (long long)(wv * OU_PASSES * G::STEPS) * 128 + lane;
    tc_f32x16 acc[RT], accx[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; accx[i][r] = 0.f; }
    const unsigned char *abase[RT];
    {
        const int r = lane & 31;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int m = 32 * i + r;
            abase[i] = lds + ((m / TW) * G::HWD + (m % TW)) * G::CELLB + (lane >> 5) * 32;
        }
    }
#pragma unroll 1
    for (int pass = 0; pass < OU_PASSES; ++pass) {
        uint4 bq[PF][2];
        const uint4 *__restrict__ wp = w2 + (long long)pass * G::STEPS * 128;
#pragma unroll
        for (int s = 0; s < PF; ++s) { bq[s][0] = wp[s * 128]; bq[s][1] = wp[s * 128 + 64]; }
        if (pass) tc_barrier();          // every wave is done with the previous pass's channels
        // ---- channels [144 pass, 144 pass + 144) of the tile (halo included) -> LDS; zeros outside the image and past channel 712
        if constexpr (!GATHER) {
            constexpr int PPC = OU_CP / 4, TOTAL = G::HCELLS * PPC, ROUNDS = (TOTAL + 511) / 512, B = 7;
            const int pc_valid = pass == OU_PASSES - 1 ? (OU_C - (OU_PASSES - 1) * OU_CP) / 4 : PPC;        // 16-byte pieces that exist in this pass
#pragma unroll 1
            for (int rr = 0; rr < ROUNDS; rr += B) {
                uint4 v[B];
#pragma unroll
                for (int k = 0; k < B; ++k) {
                    const int q = (rr + k) * 512 + tid;
                    v[k] = make_uint4(0u, 0u, 0u, 0u);
                    if (rr + k < ROUNDS && q < TOTAL) {
                        const int c = q / PPC, pc = q - c * PPC;
                        const int cy = c / G::HWD, cx = c - cy * G::HWD;
                        const int yy = y0 - 1 + cy, xx = x0 - 1 + cx;
                        if (yy >= 0 && yy < p.h && xx >= 0 && xx < p.w && pc < pc_valid)
                            v[k] = *reinterpret_cast<const uint4 *>(p.a + (img_base + (long long)yy * p.w + xx) * p.lda + pass * OU_CP + pc * 4);
                    }
                }
#pragma unroll
                for (int k = 0; k < B; ++k) {
                    const int q = (rr + k) * 512 + tid;
                    if (rr + k < ROUNDS && q < TOTAL) {
                        const int c = q / PPC, pc = q - c * PPC;
                        *reinterpret_cast<uint4 *>(lds + c * G::CELLB + pc * 16) = v[k];
                    }
                }
            }
        } else {
            // by groups of 8 channels (32 bytes of split form): group Gi = 18 pass + gl of the 90 is
            //   Gi < 32: hx[8 Gi ..] (split) | 32 <= Gi < 72: corr[8 (Gi - 32) ..] (fp32: split here) | Gi = 72: corr[320..323], flow, delta |
            //   73 <= Gi < 89: hx[256 + 8 (Gi - 73) ..] (split) | Gi = 89: the zero pad
            constexpr int GPC = OU_CP / 8, TOTAL = G::HCELLS * GPC, ROUNDS = (TOTAL + 511) / 512, B = 4;
            const float k2048l = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(0x45000000));
#pragma unroll 1
            for (int rr = 0; rr < ROUNDS; rr += B) {
                tc_f32x4 v0[B], v1[B];
                int kind[B];                                   // 0 zeros, 1 split form as is, 2 fp32 to be split
#pragma unroll
                for (int k = 0; k < B; ++k) {
                    const int q = (rr + k) * 512 + tid;
                    v0[k] = tc_f32x4{0.f, 0.f, 0.f, 0.f}; v1[k] = v0[k]; kind[k] = 0;
                    if (rr + k < ROUNDS && q < TOTAL) {
                        const int c = q / GPC, gl = q - c * GPC, Gi = GPC * pass + gl;
                        const int cy = c / G::HWD, cx = c - cy * G::HWD;
                        const int yy = y0 - 1 + cy, xx = x0 - 1 + cx;
                        if (yy >= 0 && yy < p.h && xx >= 0 && xx < p.w && Gi < 89) {
                            const long long cell = img_base + (long long)yy * p.w + xx;
                            if (Gi < 32 || Gi >= 73) {
                                const float *src = p.hx + cell * 384 + (Gi < 32 ? 8 * Gi : 256 + 8 * (Gi - 73));
                                v0[k] = *reinterpret_cast<const tc_f32x4 *>(src); v1[k] = *reinterpret_cast<const tc_f32x4 *>(src + 4); kind[k] = 1;
                            } else if (Gi < 72) {
                                const float *src = p.corr + cell * p.ld_corr + 8 * (Gi - 32);
                                v0[k] = *reinterpret_cast<const tc_f32x4 *>(src); v1[k] = *reinterpret_cast<const tc_f32x4 *>(src + 4); kind[k] = 2;
                            } else {
                                v0[k] = *reinterpret_cast<const tc_f32x4 *>(p.corr + cell * p.ld_corr + 320);
                                const float2 cc = reinterpret_cast<const float2 *>(p.coords1)[cell], dd = reinterpret_cast<const float2 *>(p.delta)[cell];
                                const float fx = cc.x - (float)xx, fy = cc.y - (float)yy;
                                v1[k] = tc_f32x4{fx, fy, dd.x, dd.y}; kind[k] = 2;
                                if (cy >= 1 && cy <= TH && cx >= 1 && cx <= TW) reinterpret_cast<float2 *>(p.flow_lr)[cell] = make_float2(fx, fy);    // the tile's own cells
                            }
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < B; ++k) {
                    const int q = (rr + k) * 512 + tid;
                    if (rr + k < ROUNDS && q < TOTAL) {
                        const int c = q / GPC, gl = q - c * GPC;
                        tc_u32x4 hi = __builtin_bit_cast(tc_u32x4, v0[k]), lo = __builtin_bit_cast(tc_u32x4, v1[k]);
                        if (kind[k] == 2) tc_split8(v0[k], v1[k], k2048l, hi, lo);
                        *reinterpret_cast<tc_u32x4 *>(lds + c * G::CELLB + gl * 32) = hi;
                        *reinterpret_cast<tc_u32x4 *>(lds + c * G::CELLB + gl * 32 + 16) = lo;
                    }
                }
            }
        }
        tc_barrier();
        tc_kloop<G, 3>(abase, wp, bq, acc, accx);
    }
    tc_barrier();           // every wave is done with the input tile: its space takes the sums

    // ---- relu(. + bias), parked [cell][256]; then the second layers as the [256 x 27] projection (as TC_RELU_PROJ)
    const float inv2048 = 1.f / 2048.f;
    float *red = reinterpret_cast<float *>(lds);
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            tc_f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][4 * b + e] + accx[i][4 * b + e] * inv2048;
            *reinterpret_cast<tc_f32x4 *>(red + (32 * i + (lane & 31)) * G::RED_ROW + 32 * wv + 8 * b + 4 * (lane >> 5)) = v;
        }
    tc_barrier();
    const float k2048 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(0x45000000));
    const int mt = wv & 3, kh = wv >> 2;
    const uint4 *__restrict__ wpj = reinterpret_cast<const uint4 *>(p.wproj) + lane;
    float *tp = reinterpret_cast<float *>(lds + G::RED_BYTES);
    if (mt < RT) {
        tc_f32x16 d, dx;
#pragma unroll
        for (int r = 0; r < 16; ++r) { d[r] = 0.f; dx[r] = 0.f; }
        const float *xrow = red + (32 * mt + (lane & 31)) * G::RED_ROW + 8 * (lane >> 5);
#pragma unroll
        for (int gg = 0; gg < 8; ++gg) {
            const int g = 8 * kh + gg;
            tc_f32x4 u = *reinterpret_cast<const tc_f32x4 *>(xrow + 16 * g), v = *reinterpret_cast<const tc_f32x4 *>(xrow + 16 * g + 4);
            u += *reinterpret_cast<const tc_f32x4 *>(p.bias + 16 * g + 8 * (lane >> 5));
            v += *reinterpret_cast<const tc_f32x4 *>(p.bias + 16 * g + 8 * (lane >> 5) + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { u[e] = relu_keep_nan(u[e]); v[e] = relu_keep_nan(v[e]); }
            tc_u32x4 hi, lo;
            tc_split8(u, v, k2048, hi, lo);
            const tc_f16x8 wh = __builtin_bit_cast(tc_f16x8, wpj[(g * 2) * 64]), wl = __builtin_bit_cast(tc_f16x8, wpj[(g * 2 + 1) * 64]);
            asm volatile("s_nop 1" : "+v"(hi), "+v"(lo));
            const tc_f16x8 xh = __builtin_bit_cast(tc_f16x8, hi), xl = __builtin_bit_cast(tc_f16x8, lo);
            d = tc_mfma(wh, xh, d);
            dx = tc_mfma(wl, xh, dx);
            dx = tc_mfma(wh, xl, dx);
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j0 = 8 * b + 4 * (lane >> 5);          // this lane's outputs j0 .. j0 + 3 of cell 32 mt + (lane & 31); 27 exist
            if (j0 < OU_PROJ_ROW) {
                tc_f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = d[4 * b + e] + dx[4 * b + e] * inv2048;
                *reinterpret_cast<tc_f32x4 *>(tp + (kh * CELLS + 32 * mt + (lane & 31)) * OU_PROJ_ROW + j0) = v;
            }
        }
    }
    tc_barrier();
    for (int idx = tid; idx < CELLS * 27; idx += 512) {
        const int m = idx / 27, j = idx - m * 27;
        const int yy = y0 + m / TW, xx = x0 + m % TW;
        if (yy < p.h && xx < p.w)
            p.tout[(img_base + (long long)yy * p.w + xx) * 27 + j] = tp[m * OU_PROJ_ROW + j] + tp[(CELLS + m) * OU_PROJ_ROW + j];
    }
}

// the first layers' weight in the GEMM's packed form [>= 256 rows][9 taps][cin_pad >= 712] fp32 -> the kernel's stream
__global__ void pack_ou_head_kernel(const float *__restrict__ wpk, int cin_pad, uint4 *__restrict__ out, long long pieces) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= pieces) return;
    constexpr int GP = OU_CP / 16, STEPS = 9 * GP;
    const int lane = (int)(idx & 63), part = (int)((idx >> 6) & 1);
    const long long t = idx >> 7;
    const int step = (int)(t % STEPS), pass = (int)((t / STEPS) % OU_PASSES), nt = (int)(t / STEPS / OU_PASSES);
    const int tap = step / GP, g = step % GP;
    const int n = 32 * nt + (lane & 31);
    unsigned w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = OU_CP * pass + 16 * g + 8 * (lane >> 5) + 2 * e;
        const long long base = ((long long)n * 9 + tap) * cin_pad + c;
        const unsigned a = split_halves(c < OU_C ? wpk[base] : 0.f), b = split_halves(c + 1 < OU_C ? wpk[base + 1] : 0.f);
        w[e] = part ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
    }
    out[idx] = make_uint4(w[0], w[1], w[2], w[3]);
}

// the second layers' weight [>= 3 rows][9 taps][256] fp32 -> MFMA A fragments of the [27 (+ 5 zero rows) x 256] matrix W2'[j = 3 tap + o][k]
__global__ void pack_proj27_kernel(const float *__restrict__ w2pk, uint4 *__restrict__ out) {
    const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (idx >= 16 * 2 * 64) return;
    const int lane = idx & 63, part = (idx >> 6) & 1, g = idx >> 7;
    const int j = lane & 31;
    unsigned w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = 16 * g + 8 * (lane >> 5) + 2 * e;
        const float v0 = j < 27 ? w2pk[((long long)(j % 3) * 9 + (j / 3)) * 256 + k] : 0.f;
        const float v1 = j < 27 ? w2pk[((long long)(j % 3) * 9 + (j / 3)) * 256 + k + 1] : 0.f;
        const unsigned a = split_halves(v0), b = split_halves(v1);
        w[e] = part ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
    }
    out[idx] = make_uint4(w[0], w[1], w[2], w[3]);
}

// out[c][o] = b2[o] + sum over the 9 taps of T[c + (dy - 1, dx - 1)][3 (3 dy + dx) + o], neighbours outside the image contributing nothing
__global__ __launch_bounds__(256) void ou_heads_sum_kernel(const float *__restrict__ T, const float *__restrict__ b2, float *__restrict__ out, int ld_out,
                                                           int P, int h, int w) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // over cells x 4 (3 used)
    if (i >= (long long)P * h * w * 4) return;
    const int o = (int)(i & 3);
    if (o == 3) return;
    const long long cell = i >> 2;
    const int rem = (int)(cell % ((long long)h * w)), y = rem / w, x = rem - y * w;
    float sum = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int yy = y + dy - 1, xx = x + dx - 1;
            if (yy >= 0 && yy < h && xx >= 0 && xx < w) sum += T[(cell + (long long)(dy - 1) * w + (dx - 1)) * 27 + 3 * (3 * dy + dx) + o];
        }
    out[cell * ld_out + o] = sum + b2[o];
}

int launch_pack_ou_head(const float *w1pk, int cin_pad, const float *w2pk, void *wtile, void *wproj, hipStream_t s) {
    if (!w1pk || !w2pk || !wtile || !wproj) return fail(MFTX_E_ARG, "pack_ou_head_weights: null pointer");
    if (cin_pad < OU_C) return fail(MFTX_E_ARG, "pack_ou_head_weights: the first layers have 712 input channels");
    if (!aligned16(wtile) || !aligned16(wproj)) return fail(MFTX_E_ALIGN, "pack_ou_head_weights: outputs not 16-byte aligned");
    const long long pieces = 8ll * OU_PASSES * 9 * (OU_CP / 16) * 128;
    hipLaunchKernelGGL(pack_ou_head_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, s, w1pk, cin_pad, reinterpret_cast<uint4 *>(wtile), pieces);
    hipLaunchKernelGGL(pack_proj27_kernel, dim3(8), dim3(256), 0, s, w2pk, reinterpret_cast<uint4 *>(wproj));
    return check_launch("pack_ou_head");
}

template <int TH, int TW, bool GATHER>
static int ou_head_launch(OuHeadArgs a, hipStream_t s) {
    using G = TcGeom<TH, TW, 3, 3, OU_CP, 256>;
    constexpr int lds_bytes = (G::A_BYTES > G::RED_BYTES ? G::A_BYTES : G::RED_BYTES) + 2 * G::CELLS * OU_PROJ_ROW * 4;
    static_assert(G::RED_BYTES + 2 * G::CELLS * OU_PROJ_ROW * 4 <= 160 * 1024 && lds_bytes <= 160 * 1024, "ou_head: LDS");
    a.tiles_x = cdiv(a.w, TW); a.tiles_y = cdiv(a.h, TH);
    const long long tiles = (long long)a.P * a.tiles_x * a.tiles_y;
    if (tiles > 0x7fffffffLL) return fail(MFTX_E_ARG, "ou_heads: too many tiles");
    auto kern = ou_head_kernel<TH, TW, GATHER>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
            return fail(MFTX_E_STATE, "ou_heads: cannot reserve %d bytes of LDS", lds_bytes);
        attr_set = true;
    }
    ProfScope prof(PC_CONV_GEMM, s, 2.0 * a.P * a.h * a.w * (256.0 * 9 * OU_C + 256.0 * 27));
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), lds_bytes, s, a);
    return check_launch("ou_heads");
}

// a: the heads' input [M][lda] in split form (712 channels), or null with its parts in `ga` (the engine); T: [M][27] scratch;
// out[M][ld_out]: occlusion logits 0, 1 and log-variance
int launch_ou_heads(const float *a, int lda, int P, int h, int w, const void *wtile, const float *b1, const void *wproj, const float *b2, float *T,
                    float *out, int ld_out, int cells, hipStream_t s, const OuGather *ga) {
    if ((!a && !ga) || !wtile || !b1 || !wproj || !b2 || !T || !out) return fail(MFTX_E_ARG, "ou_heads: null pointer");
    if (P <= 0 || h <= 0 || w <= 0 || ld_out < 3) return fail(MFTX_E_ARG, "ou_heads: bad sizes");
    if (!aligned16(wtile) || !aligned16(wproj) || !aligned16(b1) || !aligned16(T))
        return fail(MFTX_E_ALIGN, "ou_heads: weights, bias and scratch 16-byte aligned");
    OuHeadArgs k{};
    k.wf = wtile; k.bias = b1; k.wproj = wproj; k.tout = T; k.P = P; k.h = h; k.w = w;
    if (ga) {
        if (!ga->hx || !ga->corr || !ga->coords1 || !ga->delta || !ga->flow_lr) return fail(MFTX_E_ARG, "ou_heads: null pointer among the input's parts");
        if ((reinterpret_cast<uintptr_t>(ga->hx) & 31) || !aligned16(ga->corr) || ga->ld_corr % 4 || ga->ld_corr < 324 || (reinterpret_cast<uintptr_t>(ga->coords1) & 7) ||
            (reinterpret_cast<uintptr_t>(ga->delta) & 7) || (reinterpret_cast<uintptr_t>(ga->flow_lr) & 7))
            return fail(MFTX_E_ALIGN, "ou_heads: hx 32-byte, corr 16-byte (row stride a multiple of 4), coordinates 8-byte aligned");
        k.hx = ga->hx; k.corr = ga->corr; k.ld_corr = ga->ld_corr; k.coords1 = ga->coords1; k.delta = ga->delta; k.flow_lr = ga->flow_lr;
    } else {
        if ((reinterpret_cast<uintptr_t>(a) & 31) || lda % 8 || lda < OU_C) return fail(MFTX_E_ALIGN, "ou_heads: split-form rows are 32-byte aligned with strides in multiples of 8");
        k.a = a; k.lda = lda;
    }
    if (!cells) cells = tile_conv_cells(P, h, w, 3);
    int e;
    if (cells == 128) e = ga ? ou_head_launch<8, 16, true>(k, s) : ou_head_launch<8, 16, false>(k, s);
    else if (cells == 64) e = ga ? ou_head_launch<4, 16, true>(k, s) : ou_head_launch<4, 16, false>(k, s);
    else if (cells == 32) e = ga ? ou_head_launch<2, 16, true>(k, s) : ou_head_launch<2, 16, false>(k, s);
    else return fail(MFTX_E_ARG, "ou_heads: 128, 64 or 32 cells per tile");
    if (e) return e;
    const long long n = (long long)P * h * w * 4;
    ProfScope prof(PC_CONV_SMALL, s, 2.0 * P * h * w * 3 * 9);
    hipLaunchKernelGGL(ou_heads_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, T, b2, out, ld_out, P, h, w);
    return check_launch("ou_heads_sum");
}

// ---- weights: the GEMM's packed form [>= N rows][taps][cin_pad] fp32 -> [nt][ks][step][hi | lo][lane] x 16 bytes
__global__ void pack_tile_conv_kernel(const float *__restrict__ wpk, int taps, int cin, int cin_pad, int N, uint4 *__restrict__ out, long long pieces) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= pieces) return;
    const int NT = N / 32, KS = 8 / NT, GPW = cin / 16 / KS, STEPS = taps * GPW;
    const int lane = (int)(idx & 63), part = (int)((idx >> 6) & 1);
    const long long t = idx >> 7;
    const int step = (int)(t % STEPS), ks = (int)((t / STEPS) % KS), nt = (int)(t / STEPS / KS);
    const int tap = step / GPW, g = (step % GPW) * KS + ks;
    const int n = 32 * nt + (lane & 31);
    unsigned w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const long long base = ((long long)n * taps + tap) * cin_pad + 16 * g + 8 * (lane >> 5) + 2 * e;
        const unsigned a = split_halves(wpk[base]), b = split_halves(wpk[base + 1]);
        w[e] = part ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
    }
    out[idx] = make_uint4(w[0], w[1], w[2], w[3]);
}

int launch_pack_tile_conv(const float *wpk, int N, int taps, int cin, int cin_pad, void *out, hipStream_t s) {
    if (!wpk || !out) return fail(MFTX_E_ARG, "pack_tile_conv_weights: null pointer");
    if ((N != 128 && N != 256) || (cin != 128 && cin != 256) || (taps != 9 && taps != 5) || cin_pad < cin)
        return fail(MFTX_E_ARG, "pack_tile_conv_weights: N in {128, 256}, cin in {128, 256}, 5 or 9 taps");
    if (!aligned16(out)) return fail(MFTX_E_ALIGN, "pack_tile_conv_weights: output not 16-byte aligned");
    const long long pieces = (long long)N * taps * cin * 4 / 16;
    hipLaunchKernelGGL(pack_tile_conv_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, s, wpk, taps, cin, cin_pad, N,
                       reinterpret_cast<uint4 *>(out), pieces);
    return check_launch("pack_tile_conv");
}

template <int TH, int TW, int KH, int KW, int CIN, int N, int EPI>
static int tc_launch(TileConvArgs a, hipStream_t s) {
    using G = TcGeom<TH, TW, KH, KW, CIN, N>;
    a.tiles_x = cdiv(a.w, TW); a.tiles_y = cdiv(a.h, TH);
    const long long tiles = (long long)a.P * a.tiles_x * a.tiles_y;
    if (tiles > 0x7fffffffLL) return fail(MFTX_E_ARG, "tile_conv: too many tiles");
    auto kern = tile_conv_kernel<TH, TW, KH, KW, CIN, N, EPI>;
    constexpr int lds_bytes = G::LDS + (EPI == TC_RELU_PROJ ? G::PROJ_BYTES : 0);
    static_assert(lds_bytes <= 160 * 1024, "tile_conv: LDS");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
            return fail(MFTX_E_STATE, "tile_conv: cannot reserve %d bytes of LDS", lds_bytes);
        attr_set = true;
    }
    ProfScope prof(PC_CONV_GEMM, s, 2.0 * a.P * a.h * a.w * ((double)N * KH * KW * CIN + (EPI == TC_RELU_PROJ ? 256.0 * 18 : 0.0)));
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), lds_bytes, s, a);
    return check_launch("tile_conv");
}

// tile shapes by filter and cells per tile: 3 x 3: 8 x 16 / 4 x 16 / 2 x 16; 1 x 5: 4 x 32 / 2 x 32 / 1 x 32; 5 x 1: 32 x 4 / 32 x 2 / 32 x 1
template <int KH, int KW, int CIN, int N, int CELLS>
static int tc_dispatch_epi(const TileConvArgs &a, int epi, hipStream_t s) {
    constexpr int TH = KH == 3 ? CELLS / 16 : (KH == 1 ? CELLS / 32 : 32), TW = CELLS / TH;
    switch (epi) {
        case TC_LINEAR: return tc_launch<TH, TW, KH, KW, CIN, N, TC_LINEAR>(a, s);
        case TC_RELU: return tc_launch<TH, TW, KH, KW, CIN, N, TC_RELU>(a, s);
        case TC_GRU_ZR: if constexpr (N == 256 && CIN == 256) return tc_launch<TH, TW, KH, KW, CIN, N, TC_GRU_ZR>(a, s); break;
        case TC_GRU_Q: if constexpr (N == 128 && CIN == 256) return tc_launch<TH, TW, KH, KW, CIN, N, TC_GRU_Q>(a, s); break;
        case TC_RELU_PROJ: if constexpr (N == 256 && CIN == 128 && KH == 3) return tc_launch<TH, TW, KH, KW, CIN, N, TC_RELU_PROJ>(a, s); break;
    }
    return fail(MFTX_E_ARG, "tile_conv: no kernel for this epilogue and shape");
}

// which layers have a tile-resident kernel: 3 x 3 over 128 channels; 1 x 5 / 5 x 1 over 128 or 256; N = 128 or 256
bool tile_conv_applicable(int kh, int kw, int cin, int N) {
    if (N != 128 && N != 256) return false;
    if (kh == 3 && kw == 3) return cin == 128;
    if ((kh == 1 && kw == 5) || (kh == 5 && kw == 1)) return cin == 128 || cin == 256;
    return false;
}

// The kernel takes the time of ONE tile's K loop however few tiles there are (one workgroup per tile, one per CU): it pays
// only when its tiles come in nearly whole rounds of the chip (>= 5/8 full) -- 7 pairs of 64 x 64 cells are 224 tiles on 256 CUs, at
// 256 x 256 pixels (56 tiles) the ring-buffered kernel's 64 x 64 tiles are 30 % faster (bench.py, 332 vs 254 frames/s)
static int tc_num_cus() {
    static const int cus = [] {
        int dev = 0, n = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) n = prop.multiProcessorCount;
        return n;
    }();
    return cus;
}
static bool tc_fills(long long tiles) {
    const long long cus = tc_num_cus(), rounds = (tiles + cus - 1) / cus;
    return tiles * 8 >= rounds * cus * 5;      // >= 5/8: one refinement of 5 / 6 / 7 pairs of 64 x 64 cells is 2 / 8 / 7 % faster with them, of 4 pairs 8 % slower (tools/bench_pairs.py)
}
static long long tc_tiles(int P, int h, int w, int kh, int cells) {
    const int th = kh == 3 ? cells / 16 : (kh == 1 ? cells / 32 : 32), tw = cells / th;
    return (long long)P * cdiv(h, th) * cdiv(w, tw);
}
bool tile_conv_fills_chip(int P, int h, int w, int kh, int kw) { (void)kw; return tc_fills(tc_tiles(P, h, w, kh, 128)); }
bool tile_conv_small_tiles_fill(int P, int h, int w) { return tc_tiles(P, h, w, 3, 32) * 2 >= tc_num_cus(); }
// Cells per tile (round 4): 128 where such tiles fill the chip; else 64, else 32 -- a smaller tile is the same kernel with fewer
// MFMA row tiles per wave: every output is the same sequence of products and sums, THE SAME BITS, so the choice may follow the
// batch (one pair per rank of a sharded frame, a ramp-up frame, a 256 x 256 video) without a pair's result depending on it.
int tile_conv_cells(int P, int h, int w, int kh) {
    if (tc_fills(tc_tiles(P, h, w, kh, 128))) return 128;
    if (tc_fills(tc_tiles(P, h, w, kh, 64)) || tc_tiles(P, h, w, kh, 64) >= tc_num_cus()) return 64;
    return 32;
}

int launch_tile_conv(const TileConvLaunch &d, hipStream_t s) {
    if (!d.a0 || !d.wf) return fail(MFTX_E_ARG, "tile_conv: null pointer");
    if (d.P <= 0 || d.h <= 0 || d.w <= 0) return fail(MFTX_E_ARG, "tile_conv: bad sizes");
    if (!tile_conv_applicable(d.kh, d.kw, d.cin, d.N)) return fail(MFTX_E_ARG, "tile_conv: no kernel for a %d x %d convolution over %d channels, N = %d", d.kh, d.kw, d.cin, d.N);
    if (d.cin == 256 && !d.a1) return fail(MFTX_E_ARG, "tile_conv: 256 channels come as two segments of 128");
    auto bad_split = [](const float *p, int ld) { return (reinterpret_cast<uintptr_t>(p) & 31) != 0 || (ld % 8) != 0; };
    if (bad_split(d.a0, d.lda0) || (d.cin == 256 && bad_split(d.a1, d.lda1)) || !aligned16(d.wf) || (d.bias && !aligned16(d.bias)) ||
        (d.addend && (!aligned16(d.addend) || d.ld_addend % 4)))
        return fail(MFTX_E_ALIGN, "tile_conv: split-form rows are 32-byte aligned with strides in multiples of 8; weights, bias and addend 16-byte aligned");
    TileConvArgs a{};
    a.a0 = d.a0; a.lda0 = d.lda0; a.a1 = d.a1; a.lda1 = d.lda1; a.wf = d.wf; a.bias = d.bias; a.addend = d.addend; a.ld_addend = d.ld_addend;
    a.out = d.out; a.ldo = d.ldo; a.out_split = d.out_split;
    a.z = d.z; a.rh = d.rh; a.hf = d.hf; a.hx = d.hx; a.ld_hf = d.ld_hf; a.ld_hx = d.ld_hx;
    a.wproj = d.wproj; a.tout = d.tout;
    a.P = d.P; a.h = d.h; a.w = d.w;
    if (d.epi == TC_LINEAR || d.epi == TC_RELU) {
        if (!d.out || !aligned16(d.out) || d.ldo % 4 || (d.out_split && bad_split(d.out, d.ldo))) return fail(MFTX_E_ALIGN, "tile_conv: output misaligned");
    } else if (d.epi == TC_GRU_ZR) {
        if (!d.z || !d.rh || !d.hf || !aligned16(d.z) || bad_split(d.rh, 128) || !aligned16(d.hf) || d.ld_hf % 4) return fail(MFTX_E_ARG, "tile_conv: z | r epilogue operands");
    } else if (d.epi == TC_GRU_Q) {
        if (!d.z || !d.hf || !d.hx || !aligned16(d.z) || !aligned16(d.hf) || d.ld_hf % 4 || bad_split(d.hx, d.ld_hx)) return fail(MFTX_E_ARG, "tile_conv: q epilogue operands");
    } else if (d.epi == TC_RELU_PROJ) {
        if (!d.wproj || !d.tout || !d.bias || !aligned16(d.wproj) || !aligned16(d.tout)) return fail(MFTX_E_ARG, "tile_conv: projection epilogue operands");
    } else return fail(MFTX_E_ARG, "tile_conv: unknown epilogue");
    const int cells = d.cells ? d.cells : tile_conv_cells(d.P, d.h, d.w, d.kh);
    if (cells != 128 && cells != 64 && cells != 32) return fail(MFTX_E_ARG, "tile_conv: 128, 64 or 32 cells per tile");
#define TC_BY_CELLS(KH, KW, CIN, N) (cells == 128 ? tc_dispatch_epi<KH, KW, CIN, N, 128>(a, d.epi, s) : cells == 64 ? tc_dispatch_epi<KH, KW, CIN, N, 64>(a, d.epi, s) : tc_dispatch_epi<KH, KW, CIN, N, 32>(a, d.epi, s))
    if (d.kh == 3) return d.N == 256 ? TC_BY_CELLS(3, 3, 128, 256) : TC_BY_CELLS(3, 3, 128, 128);
    if (d.kh == 1) {
        if (d.cin == 128) return d.N == 256 ? TC_BY_CELLS(1, 5, 128, 256) : TC_BY_CELLS(1, 5, 128, 128);
        return d.N == 256 ? TC_BY_CELLS(1, 5, 256, 256) : TC_BY_CELLS(1, 5, 256, 128);
    }
    if (d.cin == 128) return d.N == 256 ? TC_BY_CELLS(5, 1, 128, 256) : TC_BY_CELLS(5, 1, 128, 128);
    return d.N == 256 ? TC_BY_CELLS(5, 1, 256, 256) : TC_BY_CELLS(5, 1, 256, 128);
#undef TC_BY_CELLS
}

// ---- 3 x 3 over 256 channels, tile-resident in TWO channel passes (round 6): convc2 and conv of the motion encoder -----------------
// (core/update.py:152-160: convc2 256 -> 192, conv [cor 192 | flo 64] -> 126.)  A 3 x 3 tile with its halo and 256 split-form
// channels is 187 KB -- more than a CU's LDS -- so these two layers ran on the ring-buffered kernel, re-reading their input nine
// times through L2 with a barrier per 32-wide K chunk (MFMA-busy 0.38 / 0.34, the lowest of the update block).  Here, as in
// ou_head_kernel: the tile is loaded as two passes of 128 channels (95 KB each), the accumulators live across the passes, and the K
// loop is tile_conv_kernel's -- eight ds_read_b128, two weight loads, twelve MFMAs per step, no barrier in it.
//   waves     N = 128: wave = (column tile, K half), as tile_conv_kernel.
//             N = 192: six column tiles on eight waves -- waves 0..3 own column tiles 0..3 with ALL of K, waves 4 / 5 the K halves
//             of column tile 4, waves 6 / 7 those of column tile 5: every SIMD (waves w and w + 4) carries one and a half column
//             tiles, a quarter less than the 256-wide layout would with two of its eight column tiles multiplying zeros.
//   K order   pass (channels 0..127, then 128..255) > tap > channel group: one fixed sequence of products per output, whatever
//             the batch or the tile; the halves of a K-split column tile meet in LDS and are added in a fixed order.
//   output    relu(. + bias) in split form; channels >= n_valid are left untouched (conv writes 126 of its 128: the flow sits
//             in channels 126, 127 of the motion features, core/update.py:160).
struct TileConv2pArgs {
    const float *a; int lda;            // 256 channels per cell, split form, at a + cell * lda floats
    const void *wf;                     // launch_pack_tile_conv2p
    const float *bias;                  // [n_valid]
    float *out; int ldo; int n_valid;   // split-form rows at out + cell * ldo floats
    int P, h, w, tiles_x, tiles_y;
};

// load passes + K loops + parking of one wave; G: its geometry (whole K or a K half).  A function of its own, called under the
// wave-uniform role test, so that the 128 accumulator registers of the two roles never meet in a phi (spills otherwise)
template <class G, int TH, int TW>
__device__ __forceinline__ void tc2p_main(const TileConv2pArgs &p, unsigned char *lds, const uint4 *__restrict__ w2, int ks_off, int x0, int y0, long long img_base,
                                          float *slab, int rowf, int col0) {
    constexpr int RT = G::RT, PF = 3, PASSES = 2;
    const int tid = (int)threadIdx.x, lane = tid & 63;
    tc_f32x16 acc[RT], accx[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; accx[i][r] = 0.f; }
    const unsigned char *abase[RT];
    {
        const int r = lane & 31;
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int m = 32 * i + r;
            abase[i] = lds + ((m / TW) * G::HWD + (m % TW)) * G::CELLB + (lane >> 5) * 32 + ks_off;
        }
    }
#pragma unroll 1
    for (int pass = 0; pass < PASSES; ++pass) {
        const uint4 *__restrict__ wp = w2 + (long long)pass * G::STEPS * 128;
        uint4 bq[PF][2];
#pragma unroll
        for (int s = 0; s < PF; ++s) { bq[s][0] = wp[s * 128]; bq[s][1] = wp[s * 128 + 64]; }      // in flight while the tile loads
        if (pass) tc_barrier();          // every wave is done with the first pass's channels
        // ---- channels [128 pass, 128 pass + 128) of the tile (halo included) -> LDS, 16-byte pieces, zeros outside the image
        {
            constexpr int PPC = 32, TOTAL = G::HCELLS * PPC, ROUNDS = (TOTAL + 511) / 512, B = 8;
#pragma unroll 1
            for (int r0 = 0; r0 < ROUNDS; r0 += B) {
                uint4 v[B];
#pragma unroll
                for (int k = 0; k < B; ++k) {
                    const int q = (r0 + k) * 512 + tid;
                    v[k] = make_uint4(0u, 0u, 0u, 0u);
                    if (r0 + k < ROUNDS && q < TOTAL) {
                        const int c = q / PPC, pc = q - c * PPC;
                        const int cy = c / G::HWD, cx = c - cy * G::HWD;
                        const int yy = y0 - 1 + cy, xx = x0 - 1 + cx;
                        if (yy >= 0 && yy < p.h && xx >= 0 && xx < p.w)
                            v[k] = *reinterpret_cast<const uint4 *>(p.a + (img_base + (long long)yy * p.w + xx) * p.lda + pass * 128 + pc * 4);
                    }
                }
#pragma unroll
                for (int k = 0; k < B; ++k) {
                    const int q = (r0 + k) * 512 + tid;
                    if (r0 + k < ROUNDS && q < TOTAL) {
                        const int c = q / PPC, pc = q - c * PPC;
                        *reinterpret_cast<uint4 *>(lds + c * G::CELLB + pc * 16) = v[k];
                    }
                }
            }
        }
        tc_barrier();
        tc_kloop<G, 3>(abase, wp, bq, acc, accx);
    }
    tc_barrier();           // every wave is done with the input tile: its space takes the sums
    const float inv2048 = 1.f / 2048.f;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            tc_f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][4 * b + e] + accx[i][4 * b + e] * inv2048;
            *reinterpret_cast<tc_f32x4 *>(slab + (32 * i + (lane & 31)) * rowf + col0 + 8 * b + 4 * (lane >> 5)) = v;
        }
}

template <int TH, int TW, int N>
__global__ __launch_bounds__(512, 2) void tile_conv2p_kernel(TileConv2pArgs p) {
    using GF = TcGeom<TH, TW, 3, 3, 128, 256>;      // a wave with all of K: 72 steps per pass
    using GH = TcGeom<TH, TW, 3, 3, 128, 128>;      // a wave with a K half (channel groups of its parity): 36 steps per pass
    static_assert(N == 128 || N == 192, "tile_conv2p: N");
    constexpr int CELLS = GF::CELLS, PASSES = 2;
    constexpr int RED0 = N + 4, RED1 = (N == 128 ? 128 : 64) + 4;          // floats per cell of the two slabs of parked sums
    static_assert((CELLS * (RED0 + RED1)) * 4 <= 160 * 1024, "tile_conv2p: LDS");
    extern __shared__ __attribute__((aligned(16))) unsigned char tc_lds[];
    unsigned char *lds = tc_lds;
    const int tid = (int)threadIdx.x, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = (int)blockIdx.x;
    const int tx_ = tile % p.tiles_x, ty_ = (tile / p.tiles_x) % p.tiles_y, img = tile / (p.tiles_x * p.tiles_y);
    const int x0 = tx_ * TW, y0 = ty_ * TH;
    const long long img_base = (long long)img * p.h * p.w;
    const bool full = N == 192 && wv < 4;
    const int nt = N == 128 ? (wv & 3) : (wv < 4 ? wv : 4 + ((wv - 4) >> 1));
    const int ks = N == 128 ? (wv >> 2) : (wv < 4 ? 0 : ((wv - 4) & 1));
    const int wstart = N == 128 ? wv * PASSES * GH::STEPS : (wv < 4 ? wv * PASSES * GF::STEPS : 4 * PASSES * GF::STEPS + (wv - 4) * PASSES * GH::STEPS);
    const uint4 *__restrict__ w2 = reinterpret_cast<const uint4 *>(p.wf) + (long long)wstart * 128 + (tid & 63);
    // sums -> LDS: slab 0 [cell][N] (whole-K waves and the first K halves), slab 1 (the second K halves: all columns at N = 128,
    // columns 128..191 at N = 192)
    float *red = reinterpret_cast<float *>(lds);
    const bool second = !full && ks == 1;
    float *slab = second ? red + CELLS * RED0 : red;
    const int rowf = second ? RED1 : RED0;
    const int col0 = second && N == 192 ? 32 * (nt - 4) : 32 * nt;
    if (N == 192 && full) tc2p_main<GF, TH, TW>(p, lds, w2, 0, x0, y0, img_base, slab, rowf, col0);
    else tc2p_main<GH, TH, TW>(p, lds, w2, ks * 64, x0, y0, img_base, slab, rowf, col0);
    tc_barrier();

    // ---- row-wise epilogue: 8 consecutive channels of a cell per lane
    const float k2048 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(0x45000000));
    constexpr int GPC = N / 8, ITEMS = CELLS * GPC;
#pragma unroll
    for (int it = 0; it < (ITEMS + 511) / 512; ++it) {
        const int item = tid + 512 * it;
        if (item >= ITEMS) break;
        const int m = item / GPC, n0 = (item % GPC) * 8;
        const int yy = y0 + m / TW, xx = x0 + m % TW;
        const float *src = red + m * RED0 + n0;
        tc_f32x4 u = *reinterpret_cast<const tc_f32x4 *>(src), v = *reinterpret_cast<const tc_f32x4 *>(src + 4);
        if (N == 128 || n0 >= 128) {
            const float *s1 = red + CELLS * RED0 + m * RED1 + (N == 128 ? n0 : n0 - 128);
            u += *reinterpret_cast<const tc_f32x4 *>(s1);
            v += *reinterpret_cast<const tc_f32x4 *>(s1 + 4);
        }
        if (yy >= p.h || xx >= p.w || n0 >= p.n_valid) continue;
        const long long cell = img_base + (long long)yy * p.w + xx;
        const int nv = p.n_valid - n0;                  // channels of this group that exist (>= 8: all)
        if (nv >= 8) {
            u += *reinterpret_cast<const tc_f32x4 *>(p.bias + n0);
            v += *reinterpret_cast<const tc_f32x4 *>(p.bias + n0 + 4);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) { if (e < nv) u[e] += p.bias[n0 + e]; if (4 + e < nv) v[e] += p.bias[n0 + 4 + e]; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { u[e] = relu_keep_nan(u[e]); v[e] = relu_keep_nan(v[e]); }
        tc_u32x4 hi, lo;
        tc_split8(u, v, k2048, hi, lo);
        unsigned *dst = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(p.out + cell * p.ldo) + (n0 >> 3) * 32);
        if (nv >= 8) {
            reinterpret_cast<uint4 *>(dst)[0] = __builtin_bit_cast(uint4, hi);
            reinterpret_cast<uint4 *>(dst)[1] = __builtin_bit_cast(uint4, lo);
        } else {                                        // a ragged last group: whole pairs of channels only (n_valid is even)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (2 * e + 1 < nv) { dst[e] = hi[e]; dst[4 + e] = lo[e]; }
        }
    }
}

// the GEMM's packed form [>= N rows][9 taps][cin_pad >= 256] fp32 -> the kernel's streams: wave after wave, [pass][step][hi | lo][lane] x 16 bytes
__global__ void pack_tile_conv2p_kernel(const float *__restrict__ wpk, int cin_pad, int N, uint4 *__restrict__ out, long long pieces) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= pieces) return;
    constexpr int SF = 72, SH = 36;
    const int lane = (int)(idx & 63), part = (int)((idx >> 6) & 1);
    int t = (int)(idx >> 7), wv, r;
    bool full = false;
    if (N == 128) { wv = t / (2 * SH); r = t - wv * 2 * SH; }
    else if (t < 4 * 2 * SF) { wv = t / (2 * SF); r = t - wv * 2 * SF; full = true; }
    else { t -= 4 * 2 * SF; wv = 4 + t / (2 * SH); r = t - (wv - 4) * 2 * SH; }
    const int nt = N == 128 ? (wv & 3) : (wv < 4 ? wv : 4 + ((wv - 4) >> 1));
    const int ks = N == 128 ? (wv >> 2) : (wv < 4 ? 0 : ((wv - 4) & 1));
    const int steps = full ? SF : SH, gpw = full ? 8 : 4, ksw = full ? 1 : 2;
    const int pass = r / steps, step = r - pass * steps;
    const int tap = step / gpw, g = 8 * pass + (step % gpw) * ksw + ks;
    const int n = 32 * nt + (lane & 31);
    unsigned w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const long long base = ((long long)n * 9 + tap) * cin_pad + 16 * g + 8 * (lane >> 5) + 2 * e;
        const unsigned a = split_halves(wpk[base]), b = split_halves(wpk[base + 1]);
        w[e] = part ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
    }
    out[idx] = make_uint4(w[0], w[1], w[2], w[3]);
}

int launch_pack_tile_conv2p(const float *wpk, int N, int cin_pad, void *out, hipStream_t s) {
    if (!wpk || !out) return fail(MFTX_E_ARG, "pack_tile_conv2p_weights: null pointer");
    if ((N != 128 && N != 192) || cin_pad < 256) return fail(MFTX_E_ARG, "pack_tile_conv2p_weights: N in {128, 192}, 256 input channels");
    if (!aligned16(out)) return fail(MFTX_E_ALIGN, "pack_tile_conv2p_weights: output not 16-byte aligned");
    const long long pieces = (long long)N * 9 * 256 * 4 / 16;
    hipLaunchKernelGGL(pack_tile_conv2p_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, s, wpk, cin_pad, N, reinterpret_cast<uint4 *>(out), pieces);
    return check_launch("pack_tile_conv2p");
}

template <int TH, int TW, int N>
static int tc2p_launch(TileConv2pArgs a, hipStream_t s) {
    using GF = TcGeom<TH, TW, 3, 3, 128, 256>;
    constexpr int red = GF::CELLS * ((N + 4) + ((N == 128 ? 128 : 64) + 4)) * 4;
    constexpr int lds_bytes = GF::A_BYTES > red ? GF::A_BYTES : red;
    static_assert(lds_bytes <= 160 * 1024, "tile_conv2p: LDS");
    a.tiles_x = cdiv(a.w, TW); a.tiles_y = cdiv(a.h, TH);
    const long long tiles = (long long)a.P * a.tiles_x * a.tiles_y;
    if (tiles > 0x7fffffffLL) return fail(MFTX_E_ARG, "tile_conv2p: too many tiles");
    auto kern = tile_conv2p_kernel<TH, TW, N>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
            return fail(MFTX_E_STATE, "tile_conv2p: cannot reserve %d bytes of LDS", lds_bytes);
        attr_set = true;
    }
    ProfScope prof(PC_CONV_GEMM, s, 2.0 * a.P * a.h * a.w * ((double)a.n_valid * 9 * 256));
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), lds_bytes, s, a);
    return check_launch("tile_conv2p");
}

// a [M][lda] split form, 256 channels; out [M][ldo] split form, channels [0, n_valid) written; N = 128 (n_valid <= 128, even) or 192
int launch_tile_conv2p(const float *a, int lda, const void *wf, const float *bias, float *out, int ldo, int n_valid, int P, int h, int w, int cells,
                       hipStream_t s) {
    if (!a || !wf || !bias || !out) return fail(MFTX_E_ARG, "tile_conv2p: null pointer");
    if (P <= 0 || h <= 0 || w <= 0) return fail(MFTX_E_ARG, "tile_conv2p: bad sizes");
    const int N = n_valid > 128 ? 192 : 128;
    if (n_valid < 8 || n_valid > 192 || (n_valid & 1) || (N == 192 && n_valid != 192)) return fail(MFTX_E_ARG, "tile_conv2p: 192 output channels, or an even number up to 128");
    auto bad_split = [](const float *p, int ld) { return (reinterpret_cast<uintptr_t>(p) & 31) != 0 || (ld % 8) != 0; };
    if (bad_split(a, lda) || bad_split(out, ldo) || lda < 256 || ldo < N - (N - n_valid) / 8 * 8 || !aligned16(wf) || (reinterpret_cast<uintptr_t>(bias) & 3))
        return fail(MFTX_E_ALIGN, "tile_conv2p: split-form rows are 32-byte aligned with strides in multiples of 8; weights 16-byte aligned");
    TileConv2pArgs k{};
    k.a = a; k.lda = lda; k.wf = wf; k.bias = bias; k.out = out; k.ldo = ldo; k.n_valid = n_valid; k.P = P; k.h = h; k.w = w;
    if (!cells) cells = tile_conv_cells(P, h, w, 3);
#define TC2P(TH) (N == 192 ? tc2p_launch<TH, 16, 192>(k, s) : tc2p_launch<TH, 16, 128>(k, s))
    if (cells == 128) return TC2P(8);
    if (cells == 64) return TC2P(4);
    if (cells == 32) return TC2P(2);
#undef TC2P
    return fail(MFTX_E_ARG, "tile_conv2p: 128, 64 or 32 cells per tile");
}

// ---- the flow head's second layer, in two pieces (see TC_RELU_PROJ) -----------------------------------------------------
// its filter in the GEMM's packed form [>= 2 rows][9 taps][256] fp32 -> MFMA A fragments of the [18 (+ 14 zero rows) x 256]
// matrix W2'[j = 2 tap + o][k]: [k group 0..15][hi | lo][lane] x 16 bytes
__global__ void pack_flow_head_kernel(const float *__restrict__ w2pk, uint4 *__restrict__ out) {
    const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (idx >= 16 * 2 * 64) return;
    const int lane = idx & 63, part = (idx >> 6) & 1, g = idx >> 7;
    const int j = lane & 31;
    unsigned w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = 16 * g + 8 * (lane >> 5) + 2 * e;
        const float v0 = j < 18 ? w2pk[((long long)(j & 1) * 9 + (j >> 1)) * 256 + k] : 0.f;
        const float v1 = j < 18 ? w2pk[((long long)(j & 1) * 9 + (j >> 1)) * 256 + k + 1] : 0.f;
        const unsigned a = split_halves(v0), b = split_halves(v1);
        w[e] = part ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
    }
    out[idx] = make_uint4(w[0], w[1], w[2], w[3]);
}

int launch_pack_flow_head(const float *w2pk, void *out, hipStream_t s) {
    if (!w2pk || !out) return fail(MFTX_E_ARG, "pack_flow_head_weights: null pointer");
    if (!aligned16(out)) return fail(MFTX_E_ALIGN, "pack_flow_head_weights: output not 16-byte aligned");
    hipLaunchKernelGGL(pack_flow_head_kernel, dim3(8), dim3(256), 0, s, w2pk, reinterpret_cast<uint4 *>(out));
    return check_launch("pack_flow_head");
}

// delta[c][o] = b2[o] + sum over the 9 taps of T[c + (dy - 1, dx - 1)][2 (3 dy + dx) + o], neighbours outside the image
// contributing nothing (conv2's zero padding); coords[c] += delta[c] (core/raft.py:184)
__global__ __launch_bounds__(256) void flow_head_sum_kernel(const float *__restrict__ T, const float *__restrict__ b2, float *__restrict__ delta,
                                                            const float *coords_in, float *coords_out, int P, int h, int w) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // over cells x 2
    if (i >= (long long)P * h * w * 2) return;
    const int o = (int)(i & 1);
    const long long cell = i >> 1;
    const int rem = (int)(cell % ((long long)h * w)), y = rem / w, x = rem - y * w;
    float sum = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int yy = y + dy - 1, xx = x + dx - 1;
            if (yy >= 0 && yy < h && xx >= 0 && xx < w) sum += T[(cell + (long long)(dy - 1) * w + (dx - 1)) * 18 + 2 * (3 * dy + dx) + o];
        }
    const float dl = sum + b2[o];
    delta[i] = dl;
    if (coords_out) coords_out[i] = coords_in[i] + dl;         // (in place or into another buffer: element-wise)
}

int launch_flow_head_sum(const float *T, const float *b2, float *delta, const float *coords_in, float *coords_out, int P, int h, int w, hipStream_t s) {
    if (!T || !b2 || !delta || (coords_out && !coords_in)) return fail(MFTX_E_ARG, "flow_head_sum: null pointer");
    const long long n = (long long)P * h * w * 2;
    ProfScope prof(PC_CONV_SMALL, s, 2.0 * P * h * w * 2 * 9);
    hipLaunchKernelGGL(flow_head_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, T, b2, delta, coords_in, coords_out, P, h, w);
    return check_launch("flow_head_sum");
}

}  // namespace mftx
