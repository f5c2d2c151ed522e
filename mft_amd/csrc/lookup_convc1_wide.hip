// Correlation lookup FUSED into convc1 -- the variant of round 6 that gathers the windows in SIXTEEN-BYTE PIECES (levels whose width
// is a multiple of 4: all four at 512 x 512; 40 lane addresses per cell and level instead of 100, 9-11 LDS-DMA instructions per
// unit instead of 22-25).  NOT the default: built with `make LF_WIDE=1` (-> libmftx.so) or `make lfwide` (-> ../libmftx_lfwide.so,
// what tests/test_gpu_kernels.py::test_lookup_convc1_wide_variant loads); same entry points, same results up to the order of the
// K sum (the weight stream is packed in this kernel's own K order).
//
// Why it is not the default (profiles/r6d_lookup_wide_gather.txt; same box, alternating libraries, 3 runs each): in the engine at
// 512 x 512 it runs 45.2-45.9 us against 44.0-44.2 us for csrc/lookup_convc1.hip, at 1080p 291 against 275.5 us.  The in-kernel
// timeline says why: a producer wave's step is a CHAIN -- address table (1.3 k cycles), wait for the gather issued two steps ago
// (0.8-1.2 k), tap reads (1.3 k), DMA issue + conversion (2.5 k) -- of which the DMA issue is ~0.9 k (1.9 k with dword taps); the
// wide pieces halve that, and give it back in the conversion: with the window's alignment s = x0 & 3 known only per cell, a lane
// reads its taps as 38 ds_read_b32 (19 ds_read2_b32; no unaligned ds_read_b128 on gfx950) instead of nine ds_read_b128, and the
// 656-byte cells leave room for a patch ring of two slots per wave instead of three (a slot is reused the moment its taps are in
// registers, so gathers still fly for two steps).  The gather was never the wall; the single in-order producer wave per SIMD is.
//
// What differs from csrc/lookup_convc1.hip: the patch layout (LF_WCELL), the address table (one pass of piece columns + three of
// rows), the K order (whole window rows per lane: lf_a8), the ring of two patch slots with counted vmcnt waits; the consumer waves,
// the barrier protocol and the epilogue are the same.
#include "common.h"
#include "profile.h"

namespace mftx {

typedef float lf_f32x16 __attribute__((ext_vector_type(16)));
typedef float lf_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned lf_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned lf_u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 lf_f16x8 __attribute__((ext_vector_type(8)));

constexpr int LF_GROUPS = 24;                       // 16-wide k groups: 4 levels x 6
constexpr int LF_AROW = 400;                        // bytes per A row of a unit: 96 x 4 + 16 (rows r, r + 1 start 25 sixteen-byte slots apart: conflict-free ds_read_b128)
constexpr int LF_AUNIT = 64 * LF_AROW;              // 25 600
// Window patches.  WIDE (levels whose width is a multiple of 4 -- all four at 512 x 512): a cell's 10 x 10 window is fetched as
// 10 rows x 4 SIXTEEN-BYTE pieces starting at the 4-aligned column at or below the window (round 6; 40 lane addresses per cell and
// level instead of 100): 10 rows of 16 floats, window column a at float a + s of its row, s = (window's first column) & 3; a 41st
// piece per cell that is never fetched (an out-of-range address: a zero) makes the cell stride 164 dwords, so that the conversion's
// reads of 16 cells spread over the LDS banks.  NARROW (any other level): one dword per tap, 10 rows of 10 floats, as before.
constexpr int LF_WCELL = 41 * 16;                   // 656
constexpr int LF_WROW = 64;
constexpr int LF_NCELL = 400;
constexpr int LF_NROW = 40;
constexpr int LF_PSLOT = 16 * LF_WCELL;             // 10 496 >= 16 * LF_NCELL: a unit's patches, either form
constexpr int LF_NP = 2;                            // patch ring (units) per producer wave: a unit's gather is issued one step before its conversion
constexpr int LF_CPP = 16;                          // cells per producer wave and unit (at most)
constexpr unsigned LF_WBYTES = LF_GROUPS * 4 * 4 * 1024;     // fused weights: [group][wave][fragment][lane] x 16 bytes
constexpr unsigned LF_OOB = 0x80000000u;
constexpr unsigned LF_OUT = 0x40000000u;            // table entry "outside": any sum with it is out of range
constexpr int LF_OFF_PATCH = 2 * LF_AUNIT;                                   // 51 200
constexpr int LF_OFF_STAGE = LF_OFF_PATCH + 4 * LF_NP * LF_PSLOT;            // + 83 968
constexpr int LF_OFF_COORD = LF_OFF_STAGE + 4 * 4096;                        // + 16 384
constexpr int LF_OFF_TAB = LF_OFF_COORD + 4 * 3 * 128;                       // + 1 536
constexpr int LF_LDS = LF_OFF_TAB + 4 * LF_CPP * 32 * 4;                     // + 8 192 = 161 280 bytes (of 163 840)

// K order of a level's 96 columns (shared by the conversion below and pack_lookup_convc1_kernel): the four lanes q = 0..3 that
// convert a cell own 24 columns each, k'' = 24 q + i:
//   i = 0 .. 8    sample (b = 2 q,     a = i)          b: y offset, a: x offset of the 9 x 9 samples (core/corr.py:45-51)
//   i = 9 .. 17   sample (b = 2 q + 1, a = i - 9)
//   i = 18 .. 20  sample (b = 8,       a = lf_a8(q) + i - 18) -- row 8 shared out 3 / 2 / 2 / 2; the lanes' third samples overlap
//                 their neighbours' (lf_live8): computed, finite, multiplied by a zero weight
//   i = 21 .. 23  zeros
// -- whole window rows per lane, so that every tap a lane reads lies at a compile-time offset from two per-lane addresses
// whatever the window's alignment s.
__host__ __device__ constexpr int lf_a8(int q) { return q == 0 ? 0 : q == 1 ? 3 : q == 2 ? 5 : 6; }
__host__ __device__ constexpr bool lf_live8(int q, int j) { return q == 0 || (q == 3 ? j >= 1 : j < 2); }

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
    int wide;                   // bit l: level l is gathered in 16-byte pieces (its width is a multiple of 4)
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
// necessary is always safe, so n is rounded DOWN to a multiple of 8 -- nine cases instead of 64
#define LF_W(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory"); break;
__device__ __forceinline__ void lf_wait_vmcnt(int n) {
    switch (n < 63 ? (n & ~7) : 56) {
        LF_W(0) LF_W(8) LF_W(16) LF_W(24) LF_W(32) LF_W(40) LF_W(48)
        default: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
    }
}
#undef LF_W

// one dword / sixteen bytes per lane straight into LDS, lane-linear at `dst` (wave-uniform); an out-of-range offset stores zeros
__device__ __forceinline__ void lf_dma4(__amdgpu_buffer_rsrc_t r, void *dst, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)dst, 4, voff, 0, 0, 0);
}
__device__ __forceinline__ void lf_dma16(__amdgpu_buffer_rsrc_t r, void *dst, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)dst, 16, voff, 0, 0, 0);
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
    unsigned tapn[7][4];         // narrow: table indices (row entry | column entry << 16) of this lane's tap in DMA 4 i + e of a unit
    unsigned tapw[11];           // wide: the same of this lane's piece in DMA d
    unsigned char *patches;
    float *cslots;
    unsigned *tab;               // row / column offsets of the windows being gathered: [cell][32]: rows at 0..9; columns at 16..25
                                 // (narrow) or piece columns at 10..13 (wide); [0][14] always says "outside"
    int rowc[3];                 // wide table: (cell << 8 | row j) of this lane's entry in row pass k (entries 64 k + lane of 16 x 10)
    int tcount = 0;              // (trace builds: events stamped so far)

    __device__ __forceinline__ int tile_of(int k) const { return (int)blockIdx.x + k * (int)gridDim.x; }
    __device__ __forceinline__ bool is_wide(int v) const { return (p.wide >> (v & 3)) & 1; }
    __device__ __forceinline__ bool has_coord(int v) const { return v < U && (v & 3) == 0 && (v >> 2) + 1 < my_tiles; }
    // VMEM operations of unit v's gather (gather_begin: nd) and of the coordinate prefetch behind it
    __device__ __forceinline__ int n_gather(int v) const {
#ifdef MFTX_TUNING
        if (p.ablate & 1) return 0;
#endif
        return v < U ? (is_wide(v) ? (rpw * 41 + 63) >> 6 : (((rpw + 3) & ~3) * 100 + 63) >> 6) : 0;
    }

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
    // first row / column of the window (clamped so that the int conversion is defined for wild coordinates)
    static __device__ __forceinline__ int window_origin(float sv) { return (int)fminf(fmaxf(floorf(sv), -1.0e6f), 1.0e6f) - 4; }

    // gather of unit v into patch slot v & 1.
    //   The VALU pipe of a SIMD is all a producer wave has (one instruction per four cycles), so the address arithmetic is
    // done ONCE per row and (piece) column of a window instead of once per tap: phase 1 forms, several cells per pass, the byte
    // offset of window row j (or column / piece column j) inside the query's level slice, or a large value where it lies outside the
    // level (any sum with it is out of range = a zero, as grid_sample pads), into a small LDS table; phase 2 adds one row and
    // one column entry per tap / piece.
    struct GatherCtx { __amdgpu_buffer_rsrc_t rs; unsigned char *pdst; const unsigned *tb; int nd; bool wide; };
    struct ConvCtx { float T1[30], T2[8]; float w00, w01, w10, w11; unsigned char *dst; const unsigned char *p1, *p2; bool live; };

    __device__ __forceinline__ void level_geometry(int v, const float *&base, long long &stride, unsigned &H, unsigned &W, unsigned &wb) const {
        const int l = v & 3;
        base = l == 0 ? p.lvl[0] : l == 1 ? p.lvl[1] : l == 2 ? p.lvl[2] : p.lvl[3];
        stride = l == 0 ? p.stride[0] : l == 1 ? p.stride[1] : l == 2 ? p.stride[2] : p.stride[3];
        H = (unsigned)(l == 0 ? p.hl[0] : l == 1 ? p.hl[1] : l == 2 ? p.hl[2] : p.hl[3]);
        W = (unsigned)(l == 0 ? p.wl[0] : l == 1 ? p.wl[1] : l == 2 ? p.wl[2] : p.wl[3]);
        wb = (unsigned)(l == 0 ? p.wb0 : p.wb1);
    }

    // The address table of unit v.  (All 16 cells, used or not: a DMA of a short unit runs into the next cells' entries, which must
    // say "outside" -- a stale entry could be a misaligned offset, and a misaligned dword of finite floats can be a NaN.)
    //   narrow: three cells per pass, lane = (cell of the pass, row | column, j < 10): 6 passes
    //   wide:   four cells per pass,  lane = (cell of the pass, e < 14): rows e = 0..9, piece columns e - 10 = 0..3: 4 passes
    __device__ __forceinline__ void table(int v) {
        const int k = v >> 2, l = v & 3;
        const float *base; long long stride; unsigned H, W, wb;
        level_geometry(v, base, stride, H, W, wb);
        const float inv = l == 0 ? 1.f : l == 1 ? 0.5f : l == 2 ? 0.25f : 0.125f;
        const bool blocked = l < 2;
        const unsigned rowmul = blocked ? wb * 128u : W * 4u;
        const unsigned stride4 = (unsigned)stride * 4u;
        const int cell0 = tile_of(k) * TR + pw * rpw;
        const float2 *cs = reinterpret_cast<const float2 *>(cslots + (k % 3) * 32);
        if (is_wide(v)) {
            // one pass for the piece columns -- lane = (cell, piece): all 64 lanes -- and three for the 160 rows (lane's (cell, j) of
            // pass k: rowc[k], worked out once); every pass carries ONE kind of entry, so no lane computes both and selects
            {
                const int ci = lane >> 2, j = lane & 3;
                const int o = window_origin(cs[ci].x * inv);
                const unsigned xp = (unsigned)((o & ~3) + 4 * j);                          // first cell of piece j (a multiple of 4)
                const unsigned val = blocked ? (xp >> 3) * 128u + (xp & 7u) * 4u : xp * 4u;
                const bool ok = (xp < W) & (ci < rpw) & (cell0 + ci < p.cells);            // (W is a multiple of 4: a piece is inside or outside whole)
                tab[ci * 32 + 10 + j] = ok ? val : LF_OUT;
            }
            float cy[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) cy[k] = reinterpret_cast<const float *>(cs)[2 * (rowc[k] >> 8) + 1];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int ci = rowc[k] >> 8, j = rowc[k] & 15;
                const unsigned yy = (unsigned)(window_origin(cy[k] * inv) + j);
                const unsigned val = (blocked ? __umul24(yy >> 2, rowmul) + (yy & 3u) * 32u : __umul24(yy, rowmul)) + __umul24((unsigned)ci, stride4);
                const bool ok = (yy < H) & (ci < rpw) & (cell0 + ci < p.cells);
                if (k < 2 || lane < 32) tab[ci * 32 + j] = ok ? val : LF_OUT;
            }
        } else {
            const int cl = lane / 20, rem = lane - 20 * cl;
            const int kind = rem >= 10 ? 1 : 0, j = rem - 10 * kind;
            const unsigned lim = kind ? W : H;
            float2 cv[6];
#pragma unroll
            for (int pass = 0; pass < 6; ++pass) cv[pass] = cs[min(3 * pass + cl, LF_CPP - 1)];
#pragma unroll
            for (int pass = 0; pass < 6; ++pass) {
                const int ci = 3 * pass + cl;
                const unsigned vv = (unsigned)(window_origin((kind ? cv[pass].x : cv[pass].y) * inv) + j);
                unsigned val;
                if (blocked) val = kind ? (vv >> 3) * 128u + (vv & 7u) * 4u : (vv >> 2) * rowmul + (vv & 3u) * 32u;
                else val = kind ? vv * 4u : vv * rowmul;
                if (!kind) val += (unsigned)ci * stride4;        // this cell's slice inside the unit's buffer
                const bool ok = (ci < rpw) & (cell0 + ci < p.cells) & (vv < lim);
                if (lane < 60 && ci < LF_CPP) tab[ci * 32 + kind * 16 + j] = ok ? val : LF_OUT;
            }
        }
    }

    // what unit v's DMA groups need
    __device__ __forceinline__ GatherCtx gather_begin(int v) {
        GatherCtx G;
        const float *base; long long stride; unsigned H, W, wb;
        level_geometry(v, base, stride, H, W, wb);
        const int cell0 = tile_of(v >> 2) * TR + pw * rpw;
        G.pdst = patches + (v & 1) * LF_PSLOT;          // (the slot unit v - 2 leaves the moment its taps are in registers)
        G.tb = tab;
        G.wide = is_wide(v);
        // ONE buffer descriptor per unit: this wave's cells are consecutive, their level slices lie `stride` floats apart --
        // the table has cell * stride folded into its row entries, so a tap's offset is still one addition
        const int c0 = __builtin_amdgcn_readfirstlane(cell0 < p.cells ? cell0 : 0);
        G.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base + (long long)c0 * stride), 0, (unsigned)rpw * (unsigned)stride * 4u, 0x00020000);
        // The unit's taps (pieces) form ONE array -- cell after cell, 100 (41) each -- and a DMA instruction fetches 64 consecutive
        // ones: 25 (11) instructions for 16 cells.  A lane's (cell, row, column) in DMA d never changes: their table addresses
        // were worked out once (tapn / tapw).
        G.nd = n_gather(v);
        return G;
    }

    // wide: the table entries of all 11 DMAs of a unit are read in one go, next to the conversion's tap reads (one LDS round trip
    // for both): `ow` = this lane's 11 offsets
    __device__ __forceinline__ void wide_offsets(const GatherCtx &G, unsigned (&ow)[11]) const {
#pragma unroll
        for (int d = 0; d < 11; ++d) ow[d] = G.tb[tapw[d] & 0xffffu] + G.tb[tapw[d] >> 16];
    }
    // DMAs 4 i .. 4 i + 3 of a unit's gather (narrow: their table entries are read together, so the LDS latency shows once per four)
    __device__ __forceinline__ void dma_group(const GatherCtx &G, const unsigned (&ow)[11], int i) {
        unsigned o[4];
        if (G.wide) {
            if (i > 2) return;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int d = 4 * i + e;
                if (d < 10) { if (d < G.nd) lf_dma16(G.rs, G.pdst + d * 1024, ow[d]); }
                else if (d == 10) { if (d < G.nd && lane < 16) lf_dma16(G.rs, G.pdst + d * 1024, ow[10]); }     // (16 x 41 = 10 x 64 + 16 pieces)
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = G.tb[tapn[i][e] & 0xffffu] + G.tb[tapn[i][e] >> 16];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * i + e < G.nd) lf_dma4(G.rs, G.pdst + (4 * i + e) * 256, o[e]);
        }
    }

    // conversion of unit v: lane (cell c16, q) blends its 21 samples (see lf_a8 above) and stores their halves into the A slot v & 1,
    // columns 24 q .. 24 q + 23 of the level -- in three chunks of eight columns, so that the DMAs of the next gather can be issued
    // between them (a DMA holds the wave that issues the NEXT one; VALU work slotted in between is free).  The taps: window rows
    // 2 q .. 2 q + 2 whole (30 dwords) and columns a8 .. a8 + 3 of rows 8, 9 -- 38 ds_read_b32 at compile-time offsets from two
    // per-lane addresses (wide: the window's alignment s is part of them).
    template <int ROW>
    static __device__ __forceinline__ void conv_taps(ConvCtx &C) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 10; ++c) C.T1[10 * r + c] = *reinterpret_cast<const float *>(C.p1 + r * ROW + c * 4);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) C.T2[4 * r + c] = *reinterpret_cast<const float *>(C.p2 + r * ROW + c * 4);
    }
    __device__ __forceinline__ int lf_a8x() const { return q == 0 ? 0 : q == 1 ? 3 : q == 2 ? 5 : 6; }

    // what the conversion of unit v needs apart from the taps: bilinear weights, tap addresses, destination -- worked out BEFORE
    // the wait for the unit's gather (it depends on the coordinates only)
    __device__ __forceinline__ void conv_prepare(int v, ConvCtx &C) {
        float sx, sy;
        level_coords(v, sx, sy);
        const float fx = __fsub_rn(sx, floorf(sx)), fy = __fsub_rn(sy, floorf(sy));
        const float gx = __fsub_rn(1.f, fx), gy = __fsub_rn(1.f, fy);
        C.w00 = __fmul_rn(gx, gy); C.w01 = __fmul_rn(fx, gy); C.w10 = __fmul_rn(gx, fy); C.w11 = __fmul_rn(fx, fy);
        const bool wide = is_wide(v);
        const int s = wide ? window_origin(sx) & 3 : 0;
        const unsigned char *cellp = patches + (v & 1) * LF_PSLOT + c16 * (wide ? LF_WCELL : LF_NCELL);
        const int row = wide ? LF_WROW : LF_NROW;
        C.p1 = cellp + 2 * q * row + s * 4;
        C.p2 = cellp + 8 * row + (lf_a8x() + s) * 4;
        C.dst = lds + (v & 1) * LF_AUNIT + (pw * rpw + c16) * LF_AROW + q * 96;
        C.live = c16 < rpw;
    }
    __device__ __forceinline__ void conv_load(int v, ConvCtx &C) {
        if (is_wide(v)) conv_taps<LF_WROW>(C);
        else conv_taps<LF_NROW>(C);
    }
    template <int I>
    static __device__ __forceinline__ float sample(const ConvCtx &C) {
        if constexpr (I < 18) {
            constexpr int r = I / 9, a = I - 9 * r;
            return lf_blend4(C.T1[10 * r + a], C.T1[10 * r + a + 1], C.T1[10 * r + 10 + a], C.T1[10 * r + 11 + a], C.w00, C.w01, C.w10, C.w11);
        } else if constexpr (I < 21) {
            constexpr int a = I - 18;
            return lf_blend4(C.T2[a], C.T2[a + 1], C.T2[4 + a], C.T2[5 + a], C.w00, C.w01, C.w10, C.w11);
        } else {
            return 0.f;
        }
    }
    template <int G8>
    __device__ __forceinline__ void conv_chunk(const ConvCtx &C) {
        const float k2048 = 2048.f;
        unsigned h[4], l[4];
        lf_split_pair(sample<8 * G8 + 0>(C), sample<8 * G8 + 1>(C), k2048, h[0], l[0]);
        lf_split_pair(sample<8 * G8 + 2>(C), sample<8 * G8 + 3>(C), k2048, h[1], l[1]);
        lf_split_pair(sample<8 * G8 + 4>(C), sample<8 * G8 + 5>(C), k2048, h[2], l[2]);
        lf_split_pair(sample<8 * G8 + 6>(C), sample<8 * G8 + 7>(C), k2048, h[3], l[3]);
        if (C.live) {
            *reinterpret_cast<lf_u32x4 *>(C.dst + G8 * 32) = lf_u32x4{h[0], h[1], h[2], h[3]};
            *reinterpret_cast<lf_u32x4 *>(C.dst + G8 * 32 + 16) = lf_u32x4{l[0], l[1], l[2], l[3]};
        }
    }

    // One step of a producer: the conversion of unit vc (-1: none) and the gather of unit vg = vc + 2 (-1: none) INTO THE SLOT
    // vc LEAVES: a patch slot is free the moment its taps are in the converting lanes' registers, so a ring of two slots
    // still keeps a gather in flight for two steps (round 6: the 656-byte cells of the 16-byte pieces leave room for two slots
    // per wave, not three).  Order: vg's address table (VALU + LDS) and what vc's conversion needs from the coordinates; the
    // wait for vc's patches -- counted: unit vc + 1's gather, issued a step ago, stays in flight --; vc's tap reads, complete
    // before the first DMA may overwrite them; then vg's DMA groups with a chunk of vc's conversion behind each of the first three.
    // (A producer wave is one in-order instruction stream: measured, its step is the SUM of table, DMA issue and conversion
    // whatever the interleaving -- working the table out a step ahead, pass by pass between the DMAs, was 5 % slower.)
    __device__ __forceinline__ void work(int vg, int vc) {
        GatherCtx G{};
        ConvCtx C;
#ifdef MFTX_TUNING
        if (p.ablate & 4) vc = -1;
#endif
        if (vc >= 0) conv_prepare(vc, C);        // (its coordinate read travels with the table's)
        if (vg >= 0) {
            table(vg);
            G = gather_begin(vg);
        }
        LF_T(10);
        unsigned ow[11] = {};
        if (vc >= 0) {
            // issued behind vc's gather: its coordinate prefetch, unit vc + 1's gather and that one's coordinate prefetch
            lf_wait_vmcnt((has_coord(vc) ? 1 : 0) + n_gather(vc + 1) + (has_coord(vc + 1) ? 1 : 0));
            LF_T(11);
            conv_load(vc, C);
        }
        if (vg >= 0 && G.wide) wide_offsets(G, ow);
        if (vc >= 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        LF_T(12);
#pragma unroll
        for (int i = 0; i < 7; ++i) {            // (unrolled: tapn / tapw stay in registers)
            if (vg >= 0 && 4 * i < G.nd) dma_group(G, ow, i);
            __builtin_amdgcn_sched_barrier(0);
            if (vc >= 0) {
                if (i == 0) conv_chunk<0>(C);
                if (i == 1) conv_chunk<1>(C);
                if (i == 2) conv_chunk<2>(C);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (vg >= 0 && has_coord(vg)) coords_issue((vg >> 2) + 1);
    }

    // Per step u (the consumers multiply unit u) every producer converts its cells of unit u + 1 and issues the gather of
    // unit u + 3 into the patch slot unit u + 1 leaves, interleaved (work()).
    __device__ __forceinline__ void run() {
        LF_T(1);
        if (lane == 0) tab[14] = LF_OUT;          // the entry the pad piece of every cell points at
#pragma unroll
        for (int i = 0; i < 7; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int g = 64 * (4 * i + e) + lane;                    // this lane's tap in a narrow unit's array
                const int cell = g / 100, t = g - 100 * cell, r = t / 10;
                tapn[i][e] = cell < LF_CPP ? (unsigned)(cell * 32 + r) | ((unsigned)(cell * 32 + 16 + (t - 10 * r)) << 16) : 14u | (14u << 16);
            }
#pragma unroll
        for (int d = 0; d < 11; ++d) {
            const int g = 64 * d + lane;                                  // this lane's piece in a wide unit's array
            const int cell = g / 41, t = g - 41 * cell;
            tapw[d] = (cell < LF_CPP && t < 40) ? (unsigned)(cell * 32 + (t >> 2)) | ((unsigned)(cell * 32 + 10 + (t & 3)) << 16) : 14u | (14u << 16);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int idx = min(64 * k + lane, 159);
            rowc[k] = ((idx / 10) << 8) | (idx % 10);
        }
        coords_issue(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
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
                     lane & 15, lane >> 4, {}, {},
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
// (col = lane & 31, kh = lane >> 5), the 8 halves of W[64 j + 32 jt + col][k'' = 16 (wg % 6) + 8 kh + e of level wg / 6], k'' in the
// conversion's order (lf_a8 above)
__global__ void pack_lookup_convc1_kernel(const float *__restrict__ w, int ld_w, uint4 *__restrict__ out) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;          // over 24 * 4 * 4 * 64 pieces of 16 bytes
    if (d >= LF_GROUPS * 4 * 4 * 64) return;
    const int lane = d & 63, x = (d >> 6) & 3, j = (d >> 8) & 3, wg = d >> 10;
    const int row = 64 * j + 32 * (x >> 1) + (lane & 31);
    const int lvl = wg / 6;
    _Float16 o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int kk = 16 * (wg - 6 * lvl) + 8 * (lane >> 5) + e;      // k'' = 24 q + i
        const int q = kk / 24, i = kk - 24 * q;
        int a = -1, b = 0;
        if (i < 18) { b = 2 * q + i / 9; a = i % 9; }
        else if (i < 21 && lf_live8(q, i - 18)) { b = 8; a = lf_a8(q) + i - 18; }
        float v = 0.f;
        if (a >= 0) v = w[(long long)row * ld_w + lvl * 81 + a * 9 + b];
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
    // a level is gathered in 16-byte pieces when its width is a multiple of 4 (a piece then lies inside or outside the level whole);
    // tuning builds: MFTX_LF_NARROW = 1 gathers every level tap by tap (the A/B of the gather alone)
    static const int narrow = tune_env("MFTX_LF_NARROW", 0);
    a.wide = 0;
    for (int l = 0; l < 4; ++l)
        if (!narrow && L.w[l] % 4 == 0) a.wide |= 1 << l;
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
