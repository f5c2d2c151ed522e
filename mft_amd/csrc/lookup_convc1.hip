// Correlation lookup FUSED into convc1 (core/corr.py:30-51 feeding core/update.py:152-153), split arithmetic.
//
// 323 of the lookup's 324 channels have exactly one consumer -- the 1 x 1 convolution convc1 -- 11 iterations out
// of 12.  Stand-alone, the lookup writes 37 MB of features per iteration (7 pairs of 512 x 512) that convc1 reads
// back and splits in registers.  Here the features never leave the CU: one workgroup owns a tile of up to 64 query
// cells; its four PRODUCER waves gather the cells' 10 x 10 windows (LDS-DMA, one dword per tap: what lies outside a
// level gets an out-of-range offset = grid_sample's zero padding, tap for tap), blend them, split the results into
// fp16 halves and park them in LDS in the A-operand layout of v_mfma_f32_32x32x16_f16; its four CONSUMER waves
// multiply them with convc1's weights (64 output channels per wave; the wave's weight fragments stream from L2
// straight into registers, three 16-wide k groups ahead -- no wave shares them, so LDS would add nothing) and write
// relu(. + bias) in the form convc2 reads.
//
//   unit      = (tile, pyramid level): 64 rows x 96 k -- the level's 81 window samples at k'' = 10 b + a (b: y
//               offset, a: x offset; a = 9 and k'' >= 90 carry zero weights), so that a sample's index is the index
//               of its upper-left tap in the 10 x 10 patch: every lane runs the same 24-sample loop on its own
//               quarter of a patch;
//   pipeline  = producers and consumers meet at ONE workgroup barrier per unit: barrier u certifies that unit u is
//               complete in A slot u & 1 and that the consumers are done with unit u - 1, whose slot then takes unit
//               u + 1 while the consumers multiply unit u.  Gathers run three units ahead of their conversion
//               (patch ring of three, counted vmcnt waits): the HBM round trip of a window hides under ~3 units of
//               matrix work.
//
// Results are independent of the tile size and of the batch: a row's 324 products are summed in the fixed order of
// the k groups, by one accumulator pair.
#include "common.h"
#include "profile.h"

namespace mftx {

typedef float lf_f32x16 __attribute__((ext_vector_type(16)));
typedef float lf_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned lf_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned lf_u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 lf_f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) const unsigned lf_lds_u32;

constexpr int LF_GROUPS = 24;                       // 16-wide k groups: 4 levels x 6
constexpr int LF_AROW = 400;                        // bytes per A row of a unit: 96 x 4 + 16 (rows r, r + 1 start 25 sixteen-byte slots apart: conflict-free ds_read_b128)
constexpr int LF_AUNIT = 64 * LF_AROW;              // 25 600
constexpr int LF_PATCH = 400;                       // bytes per window patch: the 100 taps of a cell and level, patches of a unit back to back
constexpr int LF_PSLOT = 16 * LF_PATCH + 32;        // a unit's patches + 8 floats that stay zero (the last cell's dummy samples read them)
constexpr int LF_NP = 3;                            // patch ring (units) per producer wave: a unit's gather is issued two steps before its conversion
constexpr int LF_CPP = 16;                          // cells per producer wave and unit (at most)
constexpr unsigned LF_WBYTES = LF_GROUPS * 4 * 4 * 1024;     // fused weights: [group][wave][fragment][lane] x 16 bytes
constexpr unsigned LF_OOB = 0x80000000u;
constexpr int LF_OFF_PATCH = 2 * LF_AUNIT;                                   // 51 200
constexpr int LF_OFF_STAGE = LF_OFF_PATCH + 4 * LF_NP * LF_PSLOT;            // + 77 184
constexpr int LF_OFF_COORD = LF_OFF_STAGE + 4 * 4096;                        // + 16 384
constexpr int LF_OFF_TAB = LF_OFF_COORD + 4 * 3 * 128;                       // + 1 536
constexpr int LF_LDS = LF_OFF_TAB + 4 * LF_CPP * 32 * 4;                     // + 8 192 = 154 496 bytes

// Tuning builds only (-DMFTX_LF_TRACE): s_memtime stamps of workgroup 0's waves at the pipeline's events, read back with
// mftx_debug_lf_trace (tools/lf_trace.py): [wave][event] = (code << 56) | ticks
#ifdef MFTX_LF_TRACE
__device__ unsigned long long lf_trace_buf[8][128];
#define LF_T(code) do { if (blockIdx.x == 0 && tcount < 128) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                        if ((threadIdx.x & 63) == 0) lf_trace_buf[threadIdx.x >> 6][tcount] = ((unsigned long long)(code) << 56) | (t_ & 0x00ffffffffffffffull); ++tcount; } } while (0)
#else
#define LF_T(code) do { } while (0)
#endif

struct LookupConvArgs {
    const float *lvl[4];
    long long stride[4];        // floats per query cell
    int hl[4], wl[4];
    int wb0, wb1;               // block-grid widths of levels 0, 1
    const float *coords;
    int cells;                  // P * h * w
    const void *wf;             // fused weights (mftx_pack_lookup_convc1_weights)
    const float *bias;
    float *out;
    int ld_out, out_split;
    int rpw;                    // cells per producer wave and tile: a tile is 4 rpw cells
    int n_tiles;
    int ablate;                 // tuning builds only (MFTX_LF_ABLATE): 1 no window gathers, 2 no MFMAs, 4 no conversion, 8 no weight loads, 16 no stores, 1024 LDS poisoned with NaNs first
};

__device__ __forceinline__ void lf_barrier() {
    // s_waitcnt lgkmcnt(0): gfx950 has back-off barriers, so the compiler inserts NO wait in front of s_barrier and the builtin is no
    // fence -- without this a wave's last ds_write may still sit in the LDS queue when another wave reads the slot behind the
    // barrier (found in round 5 with tools/race_kernels.py: harmless with the GPU to itself, wrong values under contention).
    // LDS only: global prefetches and LDS-DMA loads (vmcnt) stay in flight, their consumers count them themselves.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate).  Waiting for a smaller count than
// necessary is always safe, so n is rounded DOWN to a multiple of 8 (a gather is 8, 16, 24 or 32 operations: the exact
// counts are the ones that matter) -- nine cases instead of 64
#define LF_W(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
__device__ __forceinline__ void lf_wait_vmcnt(int n) {
    switch (n < 63 ? (n & ~7) : 56) {
        LF_W(0) LF_W(8) LF_W(16) LF_W(24) LF_W(32) LF_W(40) LF_W(48)
        default: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
    }
}
#undef LF_W

// one dword per lane straight into LDS, lane-linear at `dst` (wave-uniform); an out-of-range offset stores a zero
__device__ __forceinline__ void lf_dma4(__amdgpu_buffer_rsrc_t r, void *dst, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)dst, 4, voff, 0, 0, 0);
}

// the bilinear blend of four taps, spelled out: one multiply and three fused multiply-adds in THIS order.  Left to the compiler's
// contraction, the copies of the conversion that inlining makes (the prologue's and the loop's) may contract differently -- round 6
// saw exactly that after a code motion: 1-ulp differences between a cell converted as a workgroup's first tile and as its second,
// i.e. a pair's bits depending on its batch (tests/test_gpu_e2e.py::test_pair_bits_independent_of_batch_512, tools/lf_invariance.py).
__device__ __forceinline__ float lf_blend4(float t00, float t01, float t10, float t11, float w00, float w01, float w10, float w11) {
    return __builtin_fmaf(t11, w11, __builtin_fmaf(t10, w10, __builtin_fmaf(t01, w01, __fmul_rn(t00, w00))));
}

// (hi, lo) halves of two values: 5 instructions (conv_gemm.hip: split_pair)
__device__ __forceinline__ void lf_split_pair(float x0, float x1, float k2048, unsigned &h, unsigned &l) {
    float r0, r1;
    asm("v_cvt_pk_f16_f32 %0, %4, %5\n\t"
        "v_fma_mix_f32 %2, %0, -1.0, %4 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %3, %0, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %2, %6, 0\n\t"
        "v_fma_mixhi_f16 %1, %3, %6, 0"
        : "=&v"(h), "=&v"(l), "=&v"(r0), "=&v"(r1)
        : "v"(x0), "v"(x1), "s"(k2048));
}

// ---------------------------------------------------------------------------------------------------------------
// producer waves (pw = 0..3): cells [pw rpw, (pw + 1) rpw) of every tile
// ---------------------------------------------------------------------------------------------------------------
struct LfProducer {
    const LookupConvArgs &p;
    unsigned char *lds;
    int pw, lane, rpw, TR, my_tiles, U;
    int c16, q;
    // this lane's tap in DMA 4 i + e of a unit: the LDS addresses of its row entry and of its column entry in the table (round 6:
    // addresses, not packed indices -- the unpacking was 4 VALU instructions per DMA and lane, 100 per unit)
    lf_lds_u32 *tap_r[7][4], *tap_c[7][4];
    int ent[3];                  // (cell << 8 | j) of this lane's table entry in pass k (entries 64 k + lane of 16 x 10)
    unsigned char *patches;
    float *cslots;
    unsigned *tab;               // row / column offsets of the windows being gathered: [cell][row 0..15 | column 0..15]

    __device__ __forceinline__ int tile_of(int k) const { return (int)blockIdx.x + k * (int)gridDim.x; }

    // VMEM operations of unit v's gather, and of the coordinate prefetch that follows a level-0 unit
#ifdef MFTX_TUNING
    __device__ __forceinline__ int n_gather(int v) const { return (v < U && !(p.ablate & 1)) ? ((((rpw + 3) & ~3) * 100 + 63) >> 6) : 0; }
#else
    __device__ __forceinline__ int n_gather(int v) const { return v < U ? ((((rpw + 3) & ~3) * 100 + 63) >> 6) : 0; }
#endif
    __device__ __forceinline__ int n_coord(int v) const { return (v < U && (v & 3) == 0 && (v >> 2) + 1 < my_tiles) ? 1 : 0; }

    // coordinates of tile k's cells of this wave -> slot k % 3 (32 dwords: 16 cells x (x, y))
    __device__ __forceinline__ void coords_issue(int k) {
        const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.coords), 0, (unsigned)p.cells * 8u, 0x00020000);
        const int cell0 = __builtin_amdgcn_readfirstlane(tile_of(k) * TR + pw * rpw);
        const int cell = cell0 + (lane >> 1);
        const bool ok = (lane >> 1) < rpw && cell < p.cells;
        if (lane < 32) lf_dma4(rc, cslots + (k % 3) * 32, ok ? (unsigned)cell * 8u + (unsigned)(lane & 1) * 4u : LF_OOB);
    }

    __device__ __forceinline__ void level_coords(int v, float &sx, float &sy) const {
        const int k = v >> 2, l = v & 3;
        const float2 c = reinterpret_cast<const float2 *>(cslots + (k % 3) * 32)[c16];
        const float inv = l == 0 ? 1.f : l == 1 ? 0.5f : l == 2 ? 0.25f : 0.125f;     // (x / 2^l, exactly)
        sx = c.x * inv;
        sy = c.y * inv;
    }

    // gather of unit v into patch slot v % 3.
    //   The VALU pipe of a SIMD is all a producer wave has (one instruction per four cycles), so the address arithmetic is
    // done ONCE per row and column of a window instead of once per tap: phase 1 forms, two cells per pass -- lane =
    // (cell & 1, row | column, j) --, the byte offset of window row j (or column j) inside the query's level slice, or a
    // large value where the row / column lies outside the level (any sum with it is out of range = a zero, as
    // grid_sample pads), into a small LDS table; phase 2 adds one row and one column entry per tap.
    struct GatherCtx { __amdgpu_buffer_rsrc_t rs; unsigned char *pdst; const unsigned *tb; int nd; };
    struct ConvCtx { float T[36]; float w00, w01, w10, w11; unsigned char *dst; const unsigned char *src; bool live; };
    __device__ __forceinline__ void level_geometry(int v, const float *&base, long long &stride, unsigned &H, unsigned &W, unsigned &wb) const {
        const int l = v & 3;
        base = l == 0 ? p.lvl[0] : l == 1 ? p.lvl[1] : l == 2 ? p.lvl[2] : p.lvl[3];
        stride = l == 0 ? p.stride[0] : l == 1 ? p.stride[1] : l == 2 ? p.stride[2] : p.stride[3];
        H = (unsigned)(l == 0 ? p.hl[0] : l == 1 ? p.hl[1] : l == 2 ? p.hl[2] : p.hl[3]);
        W = (unsigned)(l == 0 ? p.wl[0] : l == 1 ? p.wl[1] : l == 2 ? p.wl[2] : p.wl[3]);
        wb = (unsigned)(l == 0 ? p.wb0 : p.wb1);
    }

    // The address table of unit v: 160 row entries and 160 column entries (16 cells x 10), as three passes of rows and three of
    // columns -- entry 64 k + lane of a kind; the lane's (cell, j) of pass k were worked out once (ent[k]).  Every pass carries
    // ONE kind of entry, so no lane computes both forms and selects, and the multiplications are 24-bit (round 6: the table was
    // 2.1 k of the producer's 6.6 k cycles per step with three cells per pass and both kinds in every pass).
    // (All 16 cells, used or not: the last DMA of a short unit runs a few taps into the next cell's table entries, which must say
    // "outside" -- a stale entry could be a misaligned offset, and a misaligned dword of finite floats can be a NaN that the
    // next cell's zero-weight dummy samples would spread over a whole row.)
    __device__ __forceinline__ void table(int v) {
        const int k = v >> 2, l = v & 3;
        const float *base; long long stride; unsigned H, W, wb;
        level_geometry(v, base, stride, H, W, wb);
        const float inv = l == 0 ? 1.f : l == 1 ? 0.5f : l == 2 ? 0.25f : 0.125f;     // (x / 2^l, exactly)
        const bool blocked = l < 2;
        const unsigned rowmul = blocked ? wb * 128u : W * 4u;
        const unsigned stride4 = (unsigned)stride * 4u;
        const int cell0 = tile_of(k) * TR + pw * rpw;
        const float *cs = cslots + (k % 3) * 32;
        float cx[3], cy[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) { cx[q] = cs[2 * (ent[q] >> 8)]; cy[q] = cs[2 * (ent[q] >> 8) + 1]; }
#pragma unroll
        for (int q = 0; q < 3; ++q) {             // rows
            const int ci = ent[q] >> 8, j = ent[q] & 15;
            // clamp so that the int conversion is defined for wild coordinates
            const unsigned yy = (unsigned)((int)fminf(fmaxf(floorf(cy[q] * inv), -1.0e6f), 1.0e6f) - 4 + j);
            const unsigned val = (blocked ? __umul24(yy >> 2, rowmul) + (yy & 3u) * 32u : __umul24(yy, rowmul)) + __umul24((unsigned)ci, stride4);
            const bool ok = (ci < rpw) & (cell0 + ci < p.cells) & (yy < H);
            if (q < 2 || lane < 32) tab[ci * 32 + j] = ok ? val : 0x40000000u;
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {             // columns
            const int ci = ent[q] >> 8, j = ent[q] & 15;
            const unsigned xx = (unsigned)((int)fminf(fmaxf(floorf(cx[q] * inv), -1.0e6f), 1.0e6f) - 4 + j);
            const unsigned val = blocked ? (xx >> 3) * 128u + (xx & 7u) * 4u : xx * 4u;
            const bool ok = (ci < rpw) & (cell0 + ci < p.cells) & (xx < W);
            if (q < 2 || lane < 32) tab[ci * 32 + 16 + j] = ok ? val : 0x40000000u;
        }
    }

    // what unit v's DMA groups need
    __device__ __forceinline__ GatherCtx gather_begin(int v) {
        GatherCtx G;
        const float *base; long long stride; unsigned H, W, wb;
        level_geometry(v, base, stride, H, W, wb);
        const int cell0 = tile_of(v >> 2) * TR + pw * rpw;
        G.pdst = patches + (v % 3) * LF_PSLOT;
        G.tb = tab;
        // ONE buffer descriptor per unit: this wave's cells are consecutive, their level slices lie `stride` floats apart --
        // the table has cell * stride folded into its row entries, so a tap's offset is still one addition
        const int c0 = __builtin_amdgcn_readfirstlane(cell0 < p.cells ? cell0 : 0);
        G.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base + (long long)c0 * stride), 0, (unsigned)rpw * (unsigned)stride * 4u, 0x00020000);
        // The unit's taps form ONE array -- cell after cell, 100 each -- and a DMA instruction fetches 64 consecutive ones:
        // 25 full instructions for 16 cells.  A lane's (cell, row, column) in DMA d never changes: their table addresses were
        // worked out once (tap_r / tap_c).  (Taps past the last cell: table entries that say "outside" -- zeros into a patch nobody reads.)
        G.nd = n_gather(v);
        return G;
    }

    // DMAs 4 i .. 4 i + 3 of a unit's gather: their table entries are read together, so the LDS latency shows once per four
    __device__ __forceinline__ void dma_group(const GatherCtx &G, int i) {
        unsigned o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = *tap_r[i][e] + *tap_c[i][e];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * i + e < G.nd) lf_dma4(G.rs, G.pdst + (4 * i + e) * 256, o[e]);
    }

    // conversion of unit v: lane (cell c16, quarter q) blends samples k'' = 24 q .. 24 q + 23 of its cell and stores
    // their halves into the A slot v & 1 -- in three chunks of eight samples, so that the DMAs of the next gather can be issued
    // between them (a DMA holds the wave that issues the NEXT one; VALU work slotted in between is free)
    // what the conversion of unit v needs apart from the taps -- bilinear weights, source, destination: worked out BEFORE the wait for
    // the unit's gather (it depends on the coordinates only; round 6: one LDS round trip less in the step's chain)
    __device__ __forceinline__ void conv_prepare(int v, ConvCtx &C) {
        float sx, sy;
        level_coords(v, sx, sy);
        const float fx = __fsub_rn(sx, floorf(sx)), fy = __fsub_rn(sy, floorf(sy));
        const float gx = __fsub_rn(1.f, fx), gy = __fsub_rn(1.f, fy);
        C.w00 = __fmul_rn(gx, gy); C.w01 = __fmul_rn(fx, gy); C.w10 = __fmul_rn(gx, fy); C.w11 = __fmul_rn(fx, fy);
        C.src = patches + (v % 3) * LF_PSLOT + c16 * LF_PATCH + q * 96;
        C.dst = lds + (v & 1) * LF_AUNIT + (pw * rpw + c16) * LF_AROW + q * 96;
        C.live = c16 < rpw;
    }
    __device__ __forceinline__ void conv_load(ConvCtx &C) {
        const lf_f32x4 *src = reinterpret_cast<const lf_f32x4 *>(C.src);
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const lf_f32x4 t = src[i];
            C.T[4 * i] = t[0]; C.T[4 * i + 1] = t[1]; C.T[4 * i + 2] = t[2]; C.T[4 * i + 3] = t[3];
        }
    }
    __device__ __forceinline__ void conv_chunk(const ConvCtx &C, int g8) {
        const float k2048 = 2048.f;
        unsigned h[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = 8 * g8 + 2 * e;
            const float v0 = lf_blend4(C.T[j], C.T[j + 1], C.T[j + 10], C.T[j + 11], C.w00, C.w01, C.w10, C.w11);
            const float v1 = lf_blend4(C.T[j + 1], C.T[j + 2], C.T[j + 11], C.T[j + 12], C.w00, C.w01, C.w10, C.w11);
            lf_split_pair(v0, v1, k2048, h[e], l[e]);
        }
        if (C.live) {
            *reinterpret_cast<lf_u32x4 *>(C.dst + g8 * 32) = lf_u32x4{h[0], h[1], h[2], h[3]};
            *reinterpret_cast<lf_u32x4 *>(C.dst + g8 * 32 + 16) = lf_u32x4{l[0], l[1], l[2], l[3]};
        }
    }

    // One step of a producer: the gather of unit vg (-1: none) and the conversion of unit vc (-1: none).  vg's address table
    // first (VALU + LDS), then the wait for vc's patches (nothing of vg is in flight yet: only unit vc + 1's gather is
    // younger), vc's patch reads, and then vg's DMA groups with a chunk of vc's conversion behind each of the first three.
    // (A producer wave is one in-order instruction stream: measured, its step is the SUM of table, DMA issue and conversion
    // whatever the interleaving -- working the table out a step ahead, pass by pass between the DMAs, was 5 % slower.)
    __device__ __forceinline__ void work(int vg, int vc) {
        GatherCtx G{};
        ConvCtx C;
        if (vg >= 0) {
            table(vg);
            G = gather_begin(vg);
        }
#ifdef MFTX_TUNING
        if (p.ablate & 1) G.nd = 0;
        if (p.ablate & 4) vc = -1;
#endif
        if (vc >= 0) {
            conv_prepare(vc, C);
            lf_wait_vmcnt(younger(vc, 1));
            conv_load(C);
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) {            // (unrolled: tap_r / tap_c stay in registers)
            if (vg >= 0 && 4 * i < G.nd) dma_group(G, i);
            __builtin_amdgcn_sched_barrier(0);
            if (vc >= 0 && i < 3) conv_chunk(C, i);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (vg >= 0 && n_coord(vg)) coords_issue((vg >> 2) + 1);
    }

    // operations issued after unit v's gather at the moment unit v is converted: the coordinate prefetch behind it and the
    // gathers of the `ahead` units issued since, with theirs
    __device__ __forceinline__ int younger(int v, int ahead) const {
        int n = n_coord(v);
        for (int a = 1; a <= ahead; ++a) n += n_gather(v + a) + n_coord(v + a);
        return n;
    }

    // Per step u (the consumers multiply unit u) every producer converts its cells of unit u + 1 and issues the gather of
    // unit u + 3 (into the patch slot unit u left a step ago), interleaved (work()).
    __device__ __forceinline__ void run() {
        int tcount = 0; (void)tcount;
        LF_T(1);
        // my patch slots start out as zeros: the dummy samples of a cell (k'' = 10 b + 9, k'' >= 90: zero weights) read up to 8
        // taps past its 100 -- the next cell's, or the slot's pad, which no DMA reaches -- and must find finite values there
        for (int i = lane; i < LF_NP * LF_PSLOT / 16; i += 64) reinterpret_cast<lf_f32x4 *>(patches)[i] = lf_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 7; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int g = 64 * (4 * i + e) + lane;                    // this lane's tap in the unit's array
                const int cell = g / 100, t = g - 100 * cell, r = t / 10;
                lf_lds_u32 *tb3 = (lf_lds_u32 *)tab;
                tap_r[i][e] = tb3 + (cell < LF_CPP ? cell * 32 + r : 0);
                tap_c[i][e] = tb3 + (cell < LF_CPP ? cell * 32 + 16 + (t - 10 * r) : 0);
            }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int idx = min(64 * q + lane, 159);
            ent[q] = ((idx / 10) << 8) | (idx % 10);
        }
        coords_issue(0);
        lf_wait_vmcnt(0);
        LF_T(2);
        work(0, -1);
        LF_T(3);
        work(1 < U ? 1 : -1, -1);
        LF_T(3);
        work(2 < U ? 2 : -1, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        LF_T(5);
        for (int u = 0; u < U; ++u) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            LF_T(6);
            lf_barrier();                    // unit u is complete in its slot; the consumers are done with unit u - 1
            LF_T(7);
            work(u + 3 < U ? u + 3 : -1, u + 1 < U ? u + 1 : -1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the conversion's patch reads and A stores are complete)
            LF_T(5);
        }
    }
};

// ---------------------------------------------------------------------------------------------------------------
// consumer waves (j = 0..3): output channels [64 j, 64 j + 64) of every tile
// ---------------------------------------------------------------------------------------------------------------
template <bool OS>
__device__ __forceinline__ void lf_consumer(const LookupConvArgs &p, unsigned char *lds, int j, int lane, int U, int TR) {
    const int col = lane & 31, kh = lane >> 5;
    const unsigned char *a_lane = lds + col * LF_AROW + kh * 32;      // + slot, + 32 it rows, + 64 g, + 16 (low halves)
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wf), 0, LF_WBYTES, 0x00020000);
    const unsigned w_lane = (unsigned)(j * 4096 + lane * 16);          // + 16384 group + 1024 fragment
    lf_f16x8 wq[3][4];                    // weight fragments of three k groups: [jt = 0 hi, lo | jt = 1 hi, lo]
    auto wload = [&](int wg, lf_f16x8 (&d)[4]) {
#ifdef MFTX_TUNING
        if (p.ablate & 8) return;
#endif
#pragma unroll
        for (int x = 0; x < 4; ++x)
            d[x] = __builtin_bit_cast(lf_f16x8, __builtin_amdgcn_raw_buffer_load_b128(rW, (unsigned)wg * 16384u + (unsigned)x * 1024u + w_lane, 0, 0));
    };
    wload(0, wq[0]); wload(1, wq[1]); wload(2, wq[2]);
    int wg_next = 3;
    // bias of this lane's four columns in the epilogue's row layout (columns 4 (lane & 7) .. + 3 of a 32-wide tile)
    lf_f32x4 bias4[2];
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) bias4[jt] = *reinterpret_cast<const lf_f32x4 *>(p.bias + 64 * j + 32 * jt + 4 * (lane & 7));
    const __amdgpu_buffer_rsrc_t rOut = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (unsigned)((long long)p.cells * p.ld_out * 4), 0x00020000);
    float *st = reinterpret_cast<float *>(lds + LF_OFF_STAGE + j * 4096);

    lf_f32x16 acc[2][2], accx[2][2];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[it][jt][r] = 0.f; accx[it][jt][r] = 0.f; }

    int tcount = 0; (void)tcount;
    LF_T(1);
    for (int u = 0; u < U; ++u) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        LF_T(6);
        lf_barrier();
        LF_T(7);
        const unsigned char *A = a_lane + (u & 1) * LF_AUNIT;
        lf_f16x8 ah[2][2], al[2][2];      // [register set][row tile]
        auto read_a = [&](int g, int set) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                ah[set][it] = *reinterpret_cast<const lf_f16x8 *>(A + it * 32 * LF_AROW + g * 64);
                al[set][it] = *reinterpret_cast<const lf_f16x8 *>(A + it * 32 * LF_AROW + g * 64 + 16);
            }
        };
        read_a(0, 0);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            const int set = g & 1;
            if (g < 5) read_a(g + 1, set ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            lf_f16x8 (&w)[4] = wq[g % 3];
#ifdef MFTX_TUNING
            if (!(p.ablate & 2))
#endif
            {
            // product by product: consecutive MFMAs never wait for each other's accumulator
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) acc[it][jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[set][it], w[2 * jt], acc[it][jt], 0, 0, 0);
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) accx[it][jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[set][it], w[2 * jt + 1], accx[it][jt], 0, 0, 0);
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) accx[it][jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[set][it], w[2 * jt], accx[it][jt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            wload(wg_next, wq[g % 3]);          // three k groups ahead of its use
            wg_next = wg_next == LF_GROUPS - 1 ? 0 : wg_next + 1;
            __builtin_amdgcn_sched_barrier(0);
        }
        LF_T(8);
        if ((u & 3) != 3) continue;
        // ---- the tile is complete: out = relu(acc + accx / 2048 + bias), through 4 KiB of the wave's own LDS so that a
        // lane holds 4 consecutive channels of a row (16-byte accesses; conv_gemm.hip's vectorised epilogue)
        const long long m_base = (long long)((int)blockIdx.x + (u >> 2) * (int)gridDim.x) * TR;
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                float *w = st + (4 * (lane >> 5)) * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    w[((r & 3) + 8 * (r >> 2)) * 32] = acc[it][jt][r] + accx[it][jt][r] * (1.f / 2048.f);
                    acc[it][jt][r] = 0.f;
                    accx[it][jt][r] = 0.f;
                }
                lf_f32x4 v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = *reinterpret_cast<const lf_f32x4 *>(st + (t * 8 + (lane >> 3)) * 32 + (lane & 7) * 4);
                const int nb = 64 * j + 32 * jt + 4 * (lane & 7);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int row = 32 * it + 8 * t + (lane >> 3);
                    const long long m = m_base + row;
#ifdef MFTX_TUNING
                    const bool ok = row < TR && m < p.cells && !(p.ablate & 16);
#else
                    const bool ok = row < TR && m < p.cells;
#endif
                    lf_f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = relu_keep_nan(v[t][e] + bias4[jt][e]);
                    if constexpr (OS) {
                        unsigned h0, h1, l0, l1;
                        const float k2048 = 2048.f;
                        lf_split_pair(o[0], o[1], k2048, h0, l0);
                        lf_split_pair(o[2], o[3], k2048, h1, l1);
                        const unsigned off = ok ? (unsigned)(m * p.ld_out * 4) + (unsigned)split_row_offset(nb) : LF_OOB;
                        __builtin_amdgcn_raw_buffer_store_b64(lf_u32x2{h0, h1}, rOut, off, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b64(lf_u32x2{l0, l1}, rOut, off + 16u, 0, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lf_u32x4, o), rOut, ok ? (unsigned)((m * p.ld_out + nb) * 4) : LF_OOB, 0, 0);
                    }
                }
            }
        LF_T(9);
    }
}

template <bool OS>
__global__ __launch_bounds__(512, 2) void lookup_convc1_kernel(LookupConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lf_lds[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#ifdef MFTX_TUNING
    if (p.ablate & 1024) {          // robustness check (tools/lf_stress.py): start from LDS full of NaNs -- nothing may depend on what it held
        for (int i = threadIdx.x; i < LF_LDS / 4; i += blockDim.x) reinterpret_cast<unsigned *>(lf_lds)[i] = 0x7fc0beefu;
        __syncthreads();
    }
#endif
    const int TR = 4 * p.rpw;
    const int my_tiles = (p.n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;    // >= 1: the grid never exceeds n_tiles
    const int U = 4 * my_tiles;
    if (wid >= 4) {
        const int pw = wid - 4;
        LfProducer P{p, lf_lds, pw, lane, p.rpw, TR, my_tiles, U,
                     lane & 15, lane >> 4, {}, {}, {},
                     lf_lds + LF_OFF_PATCH + pw * (LF_NP * LF_PSLOT),
                     reinterpret_cast<float *>(lf_lds + LF_OFF_COORD + pw * (3 * 128)),
                     reinterpret_cast<unsigned *>(lf_lds + LF_OFF_TAB + pw * (LF_CPP * 32 * 4))};
        P.run();
    } else {
        lf_consumer<OS>(p, lf_lds, wid, lane, U, TR);
    }
}

// convc1's packed fp32 weights [256][ld_w] (channel l * 81 + a * 9 + b of the lookup, core/corr.py:45-51) -> the fused
// kernel's fragment stream: for k group wg, consumer wave j, fragment x = 2 jt + (0: high, 1: low halves) and lane
// (col = lane & 31, kh = lane >> 5), the 8 halves of W[64 j + 32 jt + col][k'' = 16 (wg % 6) + 8 kh + e of level wg / 6]
__global__ void pack_lookup_convc1_kernel(const float *__restrict__ w, int ld_w, uint4 *__restrict__ out) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;          // over 24 * 4 * 4 * 64 pieces of 16 bytes
    if (d >= LF_GROUPS * 4 * 4 * 64) return;
    const int lane = d & 63, x = (d >> 6) & 3, j = (d >> 8) & 3, wg = d >> 10;
    const int row = 64 * j + 32 * (x >> 1) + (lane & 31);
    const int lvl = wg / 6;
    _Float16 o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int kk = 16 * (wg - 6 * lvl) + 8 * (lane >> 5) + e;      // k'' = 10 b + a
        const int b = kk / 10, a = kk - 10 * b;
        float v = 0.f;
        if (kk < 90 && a < 9) v = w[(long long)row * ld_w + lvl * 81 + a * 9 + b];
        const _Float16 h = (_Float16)v;
        o[e] = (x & 1) ? (_Float16)((v - (float)h) * 2048.f) : h;
    }
    uint4 r;
    __builtin_memcpy(&r, o, 16);
    out[d] = r;
}

#ifdef MFTX_LF_TRACE
extern "C" int mftx_debug_lf_trace(unsigned long long *out) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(lf_trace_buf), sizeof(unsigned long long) * 8 * 128) != hipSuccess) return -1;
    unsigned long long z[8 * 128] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(lf_trace_buf), z, sizeof z) == hipSuccess ? 0 : -1;
}
#endif

static int lf_num_cus() {
    static const int n = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        }
        return cus;
    }();
    return n;
}

int launch_pack_lookup_convc1(const float *w, int ld_w, void *out, hipStream_t s) {
    const int n = LF_GROUPS * 4 * 4 * 64;
    hipLaunchKernelGGL(pack_lookup_convc1_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, w, ld_w, reinterpret_cast<uint4 *>(out));
    return check_launch("pack_lookup_convc1");
}

bool lookup_convc1_applicable(int P, int h, int w, int ld_out) {
    const long long M = (long long)P * h * w;
    return M > 0 && M * ld_out * 4 < 0x7fffffffLL && M * 8 < 0x7fffffffLL;
}

int launch_lookup_convc1(const float *const lvl[4], const float *coords, int P, int h, int w, const void *wf,
                         const float *bias, float *out, int ld_out, int out_split, hipStream_t s) {
    LookupConvArgs a{};
    const PyramidLayout L = pyramid_layout(h, w);
    for (int l = 0; l < 4; ++l) { a.lvl[l] = lvl[l]; a.stride[l] = L.stride[l]; a.hl[l] = L.h[l]; a.wl[l] = L.w[l]; }
    a.wb0 = L.wb[0]; a.wb1 = L.wb[1];
    a.coords = coords; a.cells = P * h * w;
    a.wf = wf; a.bias = bias; a.out = out; a.ld_out = ld_out; a.out_split = out_split;
    // tile = 4 rpw cells (rpw <= 16), sized so that the tiles come in whole rounds of the CUs: 7 x 4096 cells on 256
    // CUs are 512 tiles of 56, two per CU, instead of 448 of 64 (1.75)
    const int cus = lf_num_cus();
    const long long rounds = cdiv(cdiv(a.cells, 64), cus);
    const int tr0 = cdiv(a.cells, (int)(rounds * cus));
    a.rpw = cdiv(tr0, 4) < 1 ? 1 : cdiv(tr0, 4) > LF_CPP ? LF_CPP : cdiv(tr0, 4);
    a.n_tiles = cdiv(a.cells, 4 * a.rpw);
    static const int ablate = tune_env("MFTX_LF_ABLATE", 0);
    a.ablate = ablate;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(lookup_convc1_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, LF_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(lookup_convc1_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, LF_LDS);
        if (e != hipSuccess) return fail((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    // booked as the algorithmic BYTES the fused kernel really moves: the unique taps (10 x 10 per level), the coordinates and
    // convc1's 256 output channels -- not the 324-feature tensor it no longer writes; the flops of convc1 ride along
    ProfScope prof(PC_LOOKUP_FUSED, s, (double)a.cells * (4 * 100 * 4 + 8 + 256 * 4));
    const dim3 grid(a.n_tiles < cus ? a.n_tiles : cus);
    if (out_split) hipLaunchKernelGGL(lookup_convc1_kernel<true>, grid, dim3(512), LF_LDS, s, a);
    else hipLaunchKernelGGL(lookup_convc1_kernel<false>, grid, dim3(512), LF_LDS, s, a);
    return check_launch("lookup_convc1");
}

}  // namespace mftx
