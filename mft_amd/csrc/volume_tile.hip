// All-pairs correlation volume + its three pooled levels (core/corr.py:14-28, 53-69), tile-resident form, split arithmetic.
//
// The ring-buffered GEMM (conv_gemm.hip, EPI_VOLUME) spends its time in a K loop of eight chunks -- K is only 256 -- every
// one of them exposed to the staging latency, 96 MFMAs per wave between two epilogues.  Here the 128
// target cells of a super-block (16 x 8 cells of the second feature map: four 8 x 4 blocks of the blocked level 0) stay in
// LDS for the whole workgroup (133 KB, split form, copied from the split f2 once), and every wave walks over query
// blocks of 32 cells: per 16-wide k group 8 ds_read_b128 (the four target tiles' fragments, the same addresses for every
// query block), 2 global loads of the queries' raw fp32 features (L2 -> registers, three groups ahead; split in registers
// in the shadow of the MFMAs), 12 MFMAs -- no barrier after the first, no LDS ring, no DMA.
//
// The MFMAs take the TARGETS as their first operand (D = f2_tile x f1^T, targets x queries): a lane then holds, for ONE
// query, the 4 x 4 patch (rows b, columns 4 (lane >> 5) + e) of each of the four blocks -- and the pooled levels form IN
// REGISTERS: level 1 = 2 x 2 of the patch, level 2 = 2 x 2 of those, level 3 with one exchange between the two lane halves,
// each in ATen's order ((a + b) + c + d) * 0.25 like avg_pool2d level by level (bit-identical to pooling the stored level
// 0).  No LDS staging, no barrier in the epilogue either: the eight waves drift apart and one stores while another
// multiplies.
#include "common.h"
#include "profile.h"

namespace mftx {

typedef float vt_f32x16 __attribute__((ext_vector_type(16)));
typedef float vt_f32x4 __attribute__((ext_vector_type(4)));
typedef float vt_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned vt_u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 vt_f16x8 __attribute__((ext_vector_type(8)));

constexpr int VT_C = 256;                       // feature channels
constexpr int VT_ROW = VT_C * 4 + 16;           // bytes per resident target row (consecutive targets 65 sixteen-byte slots apart)
constexpr int VT_TARGETS = 128 * VT_ROW;        // 133 120
constexpr int VT_LDS = VT_TARGETS + 8 * 1024;   // + a 1 KB transpose slab per wave (level-0 lines, see the epilogue)

struct VolTileArgs {
    const float *f1;            // [P][N][256] fp32
    const float *f2s;           // [P][N][256] split form
    float *lvl0, *lvl1, *lvl2, *lvl3;
    long long s0, s1, s2, s3;   // floats per query cell, per level (pyramid layout of common.h)
    int P, N, h, w, sbw, n_sb, wb0;
    float scale;
    int qchunks, blocks_per_chunk;      // query blocks (of 32) per workgroup
    int ablate;                         // tuning builds only (MFTX_VT_ABLATE): 1 no level-0 stores, 2 no pooled stores, 8 no level-2 / 3 stores
    int gathered;                       // pair bz's queries at f1p.p[bz] (mftx_raft_refine_gather)
    long long f2_bstride;               // floats between the pairs' target maps in f2s (0: shared)
    PairPtrs f1p;
};

__device__ __forceinline__ vt_f32x16 vt_mfma(const vt_f16x8 &a, const vt_f16x8 &b, const vt_f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// hi / lo halves of 8 consecutive k (conv_gemm.hip: split8); ends with the two wait states an MFMA needs behind a VALU write
__device__ __forceinline__ void vt_split8(const vt_f32x4 &u, const vt_f32x4 &v, float k2048, vt_f16x8 &hi, vt_f16x8 &lo) {
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
    hi = __builtin_bit_cast(vt_f16x8, vt_u32x4{h0, h1, h2, h3});
    lo = __builtin_bit_cast(vt_f16x8, vt_u32x4{l0, l1, l2, l3});
}

// lane i <- lane i + N (row_shl) / lane i - N (row_shr) inside its row of 16 lanes; lanes without a source read 0
template <int N>
__device__ __forceinline__ float vt_shl(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + N, 0xf, 0xf, true));
}
template <int N>
__device__ __forceinline__ float vt_shr(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + N, 0xf, 0xf, true));
}

// ATen's avg_pool2d order: ((a + b) + c + d) * 0.25, a b = top row, c d = bottom row
__device__ __forceinline__ float vt_pool4(float a, float b, float c, float d) { return (((a + b) + c) + d) * 0.25f; }

__global__ __launch_bounds__(512, 2) void volume_tile_kernel(VolTileArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vt_lds[];
    unsigned char *lds = vt_lds;
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wg = (int)blockIdx.x;
    const int qc = wg % p.qchunks, sb = (wg / p.qchunks) % p.n_sb, bz = wg / (p.qchunks * p.n_sb);
    const int sby = sb / p.sbw, sbx = sb - sby * p.sbw;
    const float k2048 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(0x45000000));

    // ---- the super-block's 128 target rows -> LDS: row m = 32 j + 8 y + x is cell (y, x) of block j = (block row, block column) of the 2 x 2
    {
        const float *f2p = p.f2s + (long long)bz * p.f2_bstride;
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
            const int q = k * 512 + tid, m = q >> 6, pc = q & 63;
            const int j = m >> 5, mm = m & 31;
            const int ty = 8 * sby + 4 * (j >> 1) + (mm >> 3), tx = 16 * sbx + 8 * (j & 1) + (mm & 7);
            uint4 v = make_uint4(0u, 0u, 0u, 0u);                       // targets outside the map: zero rows (their columns of the padded blocks read 0)
            if (ty < p.h && tx < p.w) v = *reinterpret_cast<const uint4 *>(f2p + ((long long)ty * p.w + tx) * VT_C + pc * 4);
            *reinterpret_cast<uint4 *>(lds + m * VT_ROW + pc * 16) = v;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (no implicit wait in front of s_barrier on gfx950: tile_conv.hip, tc_barrier)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    const unsigned char *abase = lds + (lane & 31) * VT_ROW + (lane >> 5) * 32;      // + 32 j rows, + 64 g
    const int n_blocks = (p.N + 31) >> 5;
    const int b_lo = qc * p.blocks_per_chunk, b_hi = (b_lo + p.blocks_per_chunk < n_blocks) ? b_lo + p.blocks_per_chunk : n_blocks;
    const long long qbase = (long long)bz * p.N;
    const float inv2048 = 1.f / 2048.f;
    const int hf = lane >> 5;
    const int h2 = p.h >> 2, w2 = p.w >> 2, h3 = p.h >> 3, w3 = p.w >> 3;

    // ds_bpermute byte addresses (pull): level-1 position p = 8 y1 + x1 of this half-wave <- merged lane 2 (x1 & 3) + 16 (y1 & 1) +
    // ((x1 >> 2) & 1) + 8 (y1 >> 1); level 3: the level-2 cell one row of cells down (+ 16 lanes), and its right neighbour (+ 18)
    const int pp = lane & 31;
    const int l1_src = 4 * (32 * hf + 2 * (pp & 3) + 16 * ((pp >> 3) & 1) + ((pp >> 2) & 1) + 8 * (pp >> 4));
    const int l3_src = 4 * ((lane + 16) & 63);
    constexpr int PF = 3;
    vt_f32x4 raw[PF][2];
    const float *f1base = p.gathered ? p.f1p.p[bz] : p.f1 + qbase * VT_C;
    auto row_of = [&](int qb) {
        const int q = qb * 32 + (lane & 31);
        return f1base + (long long)(q < p.N ? q : p.N - 1) * VT_C + 8 * hf;       // (rows past the last query: any row, nothing of them is stored)
    };
    auto prefetch = [&](const float *src) {
#pragma unroll
        for (int g = 0; g < PF; ++g) {
            raw[g][0] = *reinterpret_cast<const vt_f32x4 *>(src + 16 * g);
            raw[g][1] = *reinterpret_cast<const vt_f32x4 *>(src + 16 * g + 4);
        }
    };
    if (b_lo + wv < b_hi) prefetch(row_of(b_lo + wv));
    for (int qb = b_lo + wv; qb < b_hi; qb += 8) {
        const float *src = row_of(qb);
        vt_f32x16 acc[4], accx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[j][r] = 0.f; accx[j][r] = 0.f; }
        vt_f16x8 ah[2][4], al[2][4];
        auto read_a = [&](int g, int set) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ah[set][j] = *reinterpret_cast<const vt_f16x8 *>(abase + j * 32 * VT_ROW + g * 64);
                al[set][j] = *reinterpret_cast<const vt_f16x8 *>(abase + j * 32 * VT_ROW + g * 64 + 16);
            }
        };
        read_a(0, 0);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int set = g & 1;
            vt_f16x8 bh, bl;
            vt_split8(raw[g % PF][0], raw[g % PF][1], k2048, bh, bl);
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < 16) read_a(g + 1, set ^ 1);
            if (g + PF < 16) {
                raw[g % PF][0] = *reinterpret_cast<const vt_f32x4 *>(src + 16 * (g + PF));
                raw[g % PF][1] = *reinterpret_cast<const vt_f32x4 *>(src + 16 * (g + PF) + 4);
            }
            __builtin_amdgcn_sched_barrier(0);
            // (queries first: D = f1_block x f2_tile^T, queries x targets -- a lane holds ONE target cell of each block for 16 queries)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = vt_mfma(bh, ah[set][j], acc[j]);
            // (cross terms in the ring-buffered kernel's order -- query hi x target lo, then query lo x target hi: the same sequence
            // of products and sums per output, the same bits)
#pragma unroll
            for (int j = 0; j < 4; ++j) accx[j] = vt_mfma(bh, al[set][j], accx[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) accx[j] = vt_mfma(bl, ah[set][j], accx[j]);
            __builtin_amdgcn_sched_barrier(0);
        }

        // the next query block's first fragments are requested BEFORE this block's stores: memory operations of a wave complete in
        // order, loads behind 100 stores would wait for every one of them
        if (qb + 8 < b_hi) prefetch(row_of(qb + 8));
        __builtin_amdgcn_sched_barrier(0);

        // ---- epilogue, in registers.  acc[j][r]: query row 8 (r >> 2) + 4 hf + (r & 3) of the block, target cell m = lane & 31 =
        // 8 y + x of block j: per row and block a half-wave holds one whole 128-byte line of level 0.
        const int m = lane & 31;
        const int q0 = qb * 32;
        const long long blk0 = ((long long)(2 * sby) * p.wb0 + 2 * sbx) * 32 + m;          // + (j >> 1) wb0 * 32 + (j & 1) * 32
        // Levels 0 and 1 (every lane stores) leave as raw BUFFER stores: one descriptor per query block and level -- base = the block's
        // first query row, records = its valid rows, so rows past the last query fall out of range by themselves --, the lane's offset
        // inside a row in ONE 32-bit register for all stores of a level, the row in a scalar offset.  Half the address traffic of a
        // global store (64-bit address per lane) on the CU's store path, which is what this epilogue is bound by.  Levels 2 and 3
        // are global stores under EXEC (an out-of-range lane of a buffer store still takes its turn in the address unit: round 3's
        // all-buffer variant, 430 instead of 360 us) -- merged over the rows of a block since round 5, see below.
        const int rows_ok = p.N - q0 < 32 ? p.N - q0 : 32;
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(p.lvl0 + (qbase + q0) * p.s0, 0, (unsigned)((long long)rows_ok * p.s0 * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(p.lvl1 + (qbase + q0) * p.s1, 0, (unsigned)((long long)rows_ok * p.s1 * 4), 0x00020000);
        const unsigned v1off = (unsigned)((4 * hf * p.s1 + ((long long)sby * p.sbw + sbx) * 32 + m) * 4);
        const unsigned s0b = (unsigned)(p.s0 * 4), s1b = (unsigned)(p.s1 * 4), jrow = (unsigned)(p.wb0 * 128);
        // Levels 2 and 3 leave MERGED: a row of the accumulators gives 8 level-2 values and 2 level-3 values per half-wave -- as one masked
        // store each they were 32 of a query block's 112 store instructions, for 6 % of its bytes, and 11-12 % of the kernel's time
        // (MFTX_VT_ABLATE = 8: 4912 -> 4387 us at 2 x 1080p, 345 -> 303 us at 7 x 512 x 512, of which the merge recovers a quarter:
        // same box, 4921 -> 4785 / 334 -> 326 us).  Instead every row's values are pulled
        // (ds_bpermute: the lane crossbar, no LDS memory) into their place in a register that collects FOUR rows of level 2 (lane d:
        // row d >> 4 of the group, half-wave (d >> 3) & 1, cell d & 7 = 4 y + x of the super-block's 2 x 4 level-2 cells) or all SIXTEEN
        // rows of level 3 (lane d: row d >> 2, half-wave (d >> 1) & 1, cell d & 1), and a query block's levels 2 and 3 are 4 + 1 stores
        // with every lane active.  The same values: the same bits.
        int le = lane;
        asm volatile("" : "+v"(le));                // (worked out per query block, behind the K loop: nothing of it lives across the MFMAs)
        const int d2c = le & 7, d2hf = (le >> 3) & 1, d2i = le >> 4;
        const int l2_pull = 4 * (32 * d2hf + 16 * (d2c >> 2) + 2 * (d2c & 3));
        const int l2_y = 2 * sby + (d2c >> 2), l2_x = 4 * sbx + (d2c & 3);
        const bool l2_ok = l2_y < h2 && l2_x < w2;
        const int l2_off = l2_y * w2 + l2_x, l2_row = 4 * d2hf + d2i;               // + 8 (r >> 2): the query row inside the block
        const int d3c = le & 1, d3hf = (le >> 1) & 1, d3r = le >> 2;
        const int l3_pull = 4 * (32 * d3hf + 4 * d3c);
        const int l3_x = 2 * sbx + d3c;
        const bool l3_ok = sby < h3 && l3_x < w3;
        const int l3_off = sby * w3 + l3_x, l3_row = 8 * (d3r >> 2) + 4 * d3hf + (d3r & 3);
        // Level 0 leaves TRANSPOSED (round 5): per row group g (8 query rows) and block j a lane holds 4 values of ONE target column --
        // 64 dword stores per query block, two 128-byte lines each.  They go through the wave's own 1 KB of LDS instead ([8 rows][32
        // targets]: four ds_write_b32, one ds_read_b128 -- LDS serves a wave's operations in order, no barrier, no second buffer), after
        // which lane l holds targets 4 (l & 7) .. + 3 of row l >> 3: ONE dwordx4 store per (g, j) writes the eight rows' lines whole.
        // 16 stores instead of 64; the same values in the same places.  Same box, on top of the merged levels 2 / 3: 4785 -> 4722 us at
        // 2 x 1080p, 326 -> 310 us at 7 x 512 x 512.
        float *slab = reinterpret_cast<float *>(lds + VT_TARGETS + wv * 1024);
        const unsigned v0off4 = (unsigned)((le >> 3) * s0b) + (unsigned)((blk0 - m) * 4) + (unsigned)(le & 7) * 16u;
        float acc2 = 0.f, acc3 = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#ifdef MFTX_TUNING
            const bool st0g = !(p.ablate & 1);
#else
            const bool st0g = true;
#endif
            float vv[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) vv[i][j] = (acc[j][4 * g + i] + accx[j][4 * g + i] * inv2048) * p.scale;
            vt_f32x4 line[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int i = 0; i < 4; ++i) slab[(4 * hf + i) * 32 + m] = vv[i][j];
                line[j] = *reinterpret_cast<const vt_f32x4 *>(slab + (le >> 3) * 32 + (le & 7) * 4);
            }
            if (st0g) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vt_u32x4, line[j]), r0, v0off4 + (j >> 1) * jrow + (j & 1) * 128u, (unsigned)(8 * g) * s0b, 0);
            }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * g + i;
            const int qrow = q0 + 8 * (r >> 2) + 4 * hf + (r & 3);
            const bool row_ok = qrow < p.N;
#ifdef MFTX_TUNING
            const bool st1 = row_ok && !(p.ablate & 2);
#define MFTX_VT_ST23 (!(p.ablate & 10))
#else
            const bool st1 = row_ok;
#define MFTX_VT_ST23 true
#endif
            float merged = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = vv[i][j];
                // level 1: the 2 x 2 cells (y, x), (y, x + 1), (y + 1, x), (y + 1, x + 1) are lanes m, m + 1, m + 8, m + 9 of one 16-lane row
                // (valid where x and y are even), summed in ATen's order
                float t = v + vt_shl<1>(v);
                t = t + vt_shl<8>(v);
                t = (t + vt_shl<9>(v)) * 0.25f;
                // the four blocks' 8 valid lanes each -> one register, disjoint lanes: block j moves by (j & 1) + 8 (j >> 1)
                // (the shifts are pinned in front of the selects: sunk into a lane-conditional block -- the compiler does that -- a
                // DPP move finds its source lanes switched off and reads zeros)
                if (j == 0) merged = t;
                else {
                    float sh = j == 1 ? vt_shr<1>(t) : (j == 2 ? vt_shr<8>(t) : vt_shr<9>(t));
                    asm volatile("" : "+v"(sh));
                    const bool take = j == 1 ? (m & 1) != 0 : (j == 2 ? ((m >> 3) & 1) && !(m & 1) : ((m >> 3) & 1) && (m & 1));
                    merged = take ? sh : merged;
                }
            }
            // merged lane 2 xx + 16 yy + (j & 1) + 8 (j >> 1) = level-1 cell (y1, x1) = (2 (j >> 1) + yy, 4 (j & 1) + xx): pull into
            // position p = 8 y1 + x1 -- one 128-byte line of level 1 per half-wave
            const float l1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(l1_src, __builtin_bit_cast(int, merged)));
            if (st1) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, l1), r1, v1off, (unsigned)(8 * (r >> 2) + (r & 3)) * s1b, 0);
            // level 2 from the level-1 line (8 wide, 4 tall: the same lane pattern), valid at even x1, even y1: cell (y1 >> 1, x1 >> 1)
            float t2 = l1 + vt_shl<1>(l1);
            t2 = t2 + vt_shl<8>(l1);
            t2 = (t2 + vt_shl<9>(l1)) * 0.25f;
            asm volatile("" : "+v"(t2));
            {
                const float pulled = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(l2_pull, __builtin_bit_cast(int, t2)));
                acc2 = d2i == (r & 3) ? pulled : acc2;
                if ((r & 3) == 3) {
                    const int row2 = q0 + 8 * (r >> 2) + l2_row;
                    if (MFTX_VT_ST23 && l2_ok && row2 < p.N) p.lvl2[(qbase + row2) * p.s2 + l2_off] = acc2;
                }
            }
            // level 3: cells x3l = 0, 1 = level-2 cells at lanes (4 x3l, 4 x3l + 2 | 16 + 4 x3l, 16 + 4 x3l + 2)
            const float bq = vt_shl<2>(t2);
            const float cq = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(l3_src, __builtin_bit_cast(int, t2)));
            const float dq = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(l3_src + 8, __builtin_bit_cast(int, t2)));
            float v3 = (((t2 + bq) + cq) + dq) * 0.25f;
            asm volatile("" : "+v"(v3));
            {
                const float pulled = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(l3_pull, __builtin_bit_cast(int, v3)));
                acc3 = d3r == r ? pulled : acc3;
                if (r == 15) {
                    const int row3 = q0 + l3_row;
                    if (MFTX_VT_ST23 && l3_ok && row3 < p.N) p.lvl3[(qbase + row3) * p.s3 + l3_off] = acc3;
                }
            }
        }
        }
    }
}

static int vt_num_cus() {
    static const int n = [] {
        int dev = 0, cus = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        return cus;
    }();
    return n;
}

bool volume_tile_applicable(int C) { return C == VT_C; }

// f1: raw fp32 features [P][N][256]; f2s: the split form of f2 (launch_split_weights); lvl: the pyramid layout of common.h
int launch_volume_tile(const float *f1, const float *f2s, int P, int h, int w, float *const lvl[4], hipStream_t s, const PairPtrs *f1p, long long f2_bstride) {
    if (f1p && P > MFTX_MAX_GATHER) return fail(MFTX_E_ARG, "corr_pyramid: at most %d gathered pairs", MFTX_MAX_GATHER);
    const PyramidLayout L = pyramid_layout(h, w);
    VolTileArgs a{};
    a.f1 = f1; a.f2s = f2s; a.lvl0 = lvl[0]; a.lvl1 = lvl[1]; a.lvl2 = lvl[2]; a.lvl3 = lvl[3];
    a.s0 = L.stride[0]; a.s1 = L.stride[1]; a.s2 = L.stride[2]; a.s3 = L.stride[3];
    a.P = P; a.N = h * w; a.h = h; a.w = w; a.sbw = L.sbw; a.n_sb = L.sbh * L.sbw; a.wb0 = L.wb[0];
    a.scale = 1.0f / sqrtf((float)VT_C);
    a.gathered = f1p ? 1 : 0;
    if (f1p) a.f1p = *f1p;
    a.f2_bstride = f2_bstride >= 0 ? f2_bstride : (long long)a.N * VT_C;
    // One workgroup per (pair, super-block) walks over ALL query blocks when those workgroups fill at least 3/4 of the CUs (7
    // pairs of 64 x 64 cells: 224; measured 372 us against 397 with three query chunks each); smaller problems split the queries
    // into chunks, at least one query block per wave each, to put ~one workgroup on every CU.
    const int n_blocks = (a.N + 31) / 32;
    const long long base = (long long)P * a.n_sb;
    a.ablate = tune_env("MFTX_VT_ABLATE", 0);
    static const int q_forced = tune_env("MFTX_VT_QCHUNKS", 0);
    int qchunks = q_forced > 0 ? q_forced : (base * 4 >= 3LL * vt_num_cus() ? 1 : (int)((vt_num_cus() + base - 1) / base));
    if (qchunks < 1) qchunks = 1;
    if (qchunks > (n_blocks + 7) / 8) qchunks = (n_blocks + 7) / 8;
    a.blocks_per_chunk = (n_blocks + qchunks - 1) / qchunks;
    a.qchunks = (n_blocks + a.blocks_per_chunk - 1) / a.blocks_per_chunk;
    const long long wgs = base * a.qchunks;
    if (wgs > 0x7fffffffLL) return fail(MFTX_E_ARG, "corr_pyramid: too many workgroups");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(volume_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, VT_LDS) != hipSuccess)
            return fail(MFTX_E_STATE, "corr_pyramid: cannot reserve %d bytes of LDS", VT_LDS);
        attr_set = true;
    }
    ProfScope prof(PC_CORR_VOLUME, s, 2.0 * a.N * a.N * (double)VT_C * P);
    hipLaunchKernelGGL(volume_tile_kernel, dim3((unsigned)wgs), dim3(512), VT_LDS, s, a);
    return check_launch("volume_tile");
}

}  // namespace mftx
