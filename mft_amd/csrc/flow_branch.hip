// The flow branch of the motion encoder as ONE kernel: convf1 (7 x 7, 2 -> 128, ReLU) and convf2 (3 x 3, 128 -> 64, ReLU)
// on the flow = coords1 - grid (core/update.py:147-148, 154-156), split arithmetic.
//
// Stand-alone, convf1 is a thin VALU kernel (K = 98) that writes 14.7 MB of features per iteration (7 pairs of 512 x 512)
// which convf2 -- an N = 64 GEMM that cannot fill the chip's matrix pipes -- reads back nine times through L2.  Here the
// 128-channel features never leave the CU: one workgroup owns a tile of 8 x 16 cells;
//
//   stage 0   the tile's flow with a halo of 4 cells (1 for convf2 + 3 for convf1) -> LDS, zero outside the image
//             (convf1's zero padding); the flow itself goes to the tail of the GRU input (channels 382..383 of hx,
//             core/update.py:160);
//   stage 1   convf1 on the tile's 10 x 18 halo cells as a split-fp16 MFMA GEMM, M = 180 (six 32-row tiles), N = 128,
//             K = 7 filter rows x 16 (7 taps x 2 channels + 2 zero slots): a lane's 8 k of an A fragment are 8 consecutive
//             floats of the flow tile; the wave's weight fragments (64 channels) live in registers.  relu(. + bias), ZERO
//             for halo cells outside the image (convf2's zero padding), split into fp16 halves -> LDS, 528 bytes per cell;
//   stage 2   convf2 from that LDS tile: M = 128, N = 64, K = 9 taps x 128.  Wave (nt, kq) owns output channels
//             [32 nt, 32 nt + 32) of all 128 cells and a quarter of K (channel groups 2 kq, 2 kq + 1 of every tap): its
//             weight fragments stream from L2 straight into registers (no other wave of the workgroup reads them), the A
//             fragments are ds_read_b128 at compile-time offsets;
//   stage 3   the four K quarters meet in LDS, are summed in a fixed order, relu(. + bias), and leave as the 64 flow
//             channels of `corflo` in split form, 32 contiguous bytes per lane.
//
// Results are independent of the batch and of where a cell lies in its tile: every output is the same sequence of
// products and sums.  Operands are split as in the GEMMs (x = hi + lo / 2048): a flow beyond the fp16 range (65504 px)
// gives NaN, never a finite wrong value.
#include "common.h"
#include "profile.h"

namespace mftx {

typedef float fb_f32x16 __attribute__((ext_vector_type(16)));
typedef float fb_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned fb_u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 fb_f16x8 __attribute__((ext_vector_type(8)));

constexpr int FB_TH = 8, FB_TW = 16;                 // output tile (cells)
constexpr int FB_HW = FB_TW + 2;                     // halo tile: 10 x 18 cells
constexpr int FB_HCELLS = (FB_TH + 2) * FB_HW;       // 180
constexpr int FB_CELL = 528;                         // bytes per halo cell: 128 channels in split form + 16 (consecutive cells start 33 sixteen-byte slots apart)
constexpr int FB_FCOLS = 26;                         // flow tile: 16 rows x 26 cells x (fx, fy); column 24 feeds the zero-weight slot, 25 pads
constexpr int FB_FROW = 28;                          // cells per row of a shifted copy of the split flow tile (112 bytes: 16-byte aligned rows)
constexpr int FB_OFF_FLOW = 192 * FB_CELL;           // 101 376 (six whole row tiles of stage 1)
constexpr int FB_FLOW_BYTES = 4 * 2 * 16 * FB_FROW * 4;     // [shift 0..3][hi | lo][row][cell] x (fx | fy << 16) fp16 pairs: 14 336
constexpr int FB_RED_ROW = 68;                       // floats per cell of a K quarter's partial sums (64 + 4)
constexpr int FB_LDS = 4 * 128 * FB_RED_ROW * 4;     // 139 264: stage 3's partial sums reuse everything
static_assert(FB_OFF_FLOW + FB_FLOW_BYTES <= FB_LDS, "stage 3 is the largest user of LDS");
constexpr unsigned FB_W1_BYTES = 7 * 4 * 2 * 1024;           // [filter row][column tile][hi | lo][lane] x 16 bytes
constexpr unsigned FB_W2_BYTES = 2 * 4 * 18 * 2 * 1024;      // [nt][kq][step][hi | lo][lane] x 16 bytes
constexpr unsigned FB_WBYTES = FB_W1_BYTES + FB_W2_BYTES;    // 352 256

// Tuning builds only (-DMFTX_LF_TRACE): s_memtime stamps of workgroup 0's waves at the stage boundaries, read back with
// mftx_debug_fb_trace (tools/fb_trace.py): [wave][event] = (code << 56) | ticks
#ifdef MFTX_LF_TRACE
__device__ unsigned long long fb_trace_buf[8][16];
#define FB_T(code) do { if (blockIdx.x == 0 && tcount < 16) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                        if ((threadIdx.x & 63) == 0) fb_trace_buf[threadIdx.x >> 6][tcount] = ((unsigned long long)(code) << 56) | (t_ & 0x00ffffffffffffffull); ++tcount; } } while (0)
#else
#define FB_T(code) do { } while (0)
#endif

struct FlowBranchArgs {
    const float *coords;        // [P, h, w, 2]
    const void *wf;             // mftx_pack_flow_branch_weights
    const float *b1, *b2;
    float *out; int ld_out;     // 64 channels per cell in split form at out + cell * ld_out (floats)
    float *hx; int ld_hx;       // split-form GRU input: the flow goes to channels 382, 383 (nullptr: not written)
    // Optional: the previous iteration's flow-head update, not applied yet.  T [P*h*w][18] = the head's partial products
    // (tile_conv.hip: TC_RELU_PROJ), b2h its bias: every cell of the tile (halo included) first becomes coords + delta,
    // delta[o] = b2h[o] + the nine shifted T terms -- exactly flow_head_sum_kernel's sum -- and the tile's own cells are
    // written to coords_out (ANOTHER buffer: the halo cells of this tile are other workgroups' own cells) and delta_out.
    const float *T, *b2h;
    float *coords_out, *delta_out;
    int P, h, w, tiles_x, tiles_y;
};

__device__ __forceinline__ void fb_barrier() {
    // s_waitcnt lgkmcnt(0): gfx950 has back-off barriers, so the compiler inserts NO wait in front of s_barrier and the builtin is no
    // fence -- without this a wave's last ds_write may still sit in the LDS queue when another wave reads the slot behind the
    // barrier (found in round 5 with tools/race_kernels.py: harmless with the GPU to itself, wrong values under contention).
    // LDS only: global prefetches and LDS-DMA loads (vmcnt) stay in flight, their consumers count them themselves.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// hi / lo halves of 8 consecutive k (conv_gemm.hip: split8)
__device__ __forceinline__ void fb_split8(const fb_f32x4 &u, const fb_f32x4 &v, float k2048, fb_f16x8 &hi, fb_f16x8 &lo) {
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
    hi = __builtin_bit_cast(fb_f16x8, fb_u32x4{h0, h1, h2, h3});
    lo = __builtin_bit_cast(fb_f16x8, fb_u32x4{l0, l1, l2, l3});
}

// the same for two values (conv_gemm.hip: split_pair)
__device__ __forceinline__ void fb_split_pair(float x0, float x1, float k2048, unsigned &h, unsigned &l) {
    float r0, r1;
    asm("v_cvt_pk_f16_f32 %0, %4, %5\n\t"
        "v_fma_mix_f32 %2, %0, -1.0, %4 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %3, %0, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %2, %6, 0\n\t"
        "v_fma_mixhi_f16 %1, %3, %6, 0"
        : "=&v"(h), "=&v"(l), "=&v"(r0), "=&v"(r1)
        : "v"(x0), "v"(x1), "s"(k2048));
}

__device__ __forceinline__ fb_f32x16 fb_mfma(const fb_f16x8 &a, const fb_f16x8 &b, const fb_f32x16 &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(512, 2) void flow_branch_kernel(FlowBranchArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fb_lds[];
    unsigned char *lds = fb_lds;
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = (int)blockIdx.x;
    const int tx_ = tile % p.tiles_x, ty_ = (tile / p.tiles_x) % p.tiles_y, img = tile / (p.tiles_x * p.tiles_y);
    const int x0 = tx_ * FB_TW, y0 = ty_ * FB_TH;
    const long long img_base = (long long)img * p.h * p.w;
    const float k2048 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(0x45000000));
    const uint4 *__restrict__ wf = reinterpret_cast<const uint4 *>(p.wf);
#ifdef MFTX_LF_TRACE
    int tcount = 0;
#endif
    FB_T(1);

    // ---- stage 1 weights: this wave's 32 channels (column tile j1), all 7 filter rows -> registers
    const int j1 = wv & 3;
    fb_f16x8 w1h[7], w1l[7];
#pragma unroll
    for (int g = 0; g < 7; ++g) {
        const uint4 *src = wf + ((g * 4 + j1) * 2) * 64 + lane;
        w1h[g] = __builtin_bit_cast(fb_f16x8, src[0]);
        w1l[g] = __builtin_bit_cast(fb_f16x8, src[64]);
    }
    // (stage 1 computes the TRANSPOSED product, channels x cells: a lane ends up with 4 x 4 consecutive channels of ONE cell --
    // channels 32 j1 + 8 b + 4 (lane >> 5) + 0..3 for b = 0..3)
    fb_f32x4 bias1[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) bias1[b] = *reinterpret_cast<const fb_f32x4 *>(p.b1 + 32 * j1 + 8 * b + 4 * (lane >> 5));

    // ---- stage 0: flow tile -> LDS
    // The flow is split ONCE, here: cell (r, c) -> (hi_x | hi_y << 16) and (lo_x | lo_y << 16).  An A fragment of stage 1 is
    // four consecutive cells of a row starting at an arbitrary cell; ds_read_b128 wants 16-byte alignment, so the tile is kept
    // as four copies shifted by 0..3 cells: a fragment starting at cell c is aligned in copy c & 3.
    unsigned *fsp = reinterpret_cast<unsigned *>(lds + FB_OFF_FLOW);
    if (tid < 16 * FB_FCOLS) {
        const int r = tid / FB_FCOLS, c = tid - r * FB_FCOLS;
        const int yy = y0 - 4 + r, xx = x0 - 4 + c;
        float2 f = make_float2(0.f, 0.f);
        unsigned hw = 0u, lw = 0u;
        if (yy >= 0 && yy < p.h && xx >= 0 && xx < p.w) {
            const long long cell = img_base + (long long)yy * p.w + xx;
            float2 cd = *reinterpret_cast<const float2 *>(p.coords + 2 * cell);
            if (p.T) {                                                   // (flow_head_sum_kernel's arithmetic, term by term)
                float sx = 0.f, sy = 0.f;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int ny = yy + dy - 1, nx = xx + dx - 1;
                        if (ny >= 0 && ny < p.h && nx >= 0 && nx < p.w) {
                            const float2 t = *reinterpret_cast<const float2 *>(p.T + (cell + (long long)(dy - 1) * p.w + (dx - 1)) * 18 + 2 * (3 * dy + dx));
                            sx += t.x;
                            sy += t.y;
                        }
                    }
                const float dlx = sx + p.b2h[0], dly = sy + p.b2h[1];
                cd.x += dlx;
                cd.y += dly;
                if (r >= 4 && r < 4 + FB_TH && c >= 4 && c < 4 + FB_TW) {
                    *reinterpret_cast<float2 *>(p.coords_out + 2 * cell) = cd;
                    *reinterpret_cast<float2 *>(p.delta_out + 2 * cell) = make_float2(dlx, dly);
                }
            }
            f.x = cd.x - (float)xx;
            f.y = cd.y - (float)yy;
            const unsigned sx = split_halves(f.x), sy = split_halves(f.y);
            hw = (sx & 0xffffu) | (sy << 16);
            lw = (sx >> 16) | (sy & 0xffff0000u);
            if (p.hx && r >= 4 && r < 4 + FB_TH && c >= 4 && c < 4 + FB_TW) {      // the tile's own cells: flow -> hx[382..383]
                char *dst = reinterpret_cast<char *>(p.hx + cell * p.ld_hx) + split_row_offset(382);
                *reinterpret_cast<unsigned *>(dst) = hw;
                *reinterpret_cast<unsigned *>(dst + 16) = lw;
            }
        }
#pragma unroll
        for (int sft = 0; sft < 4; ++sft) {
            if (c >= sft) {
                fsp[((sft * 2 + 0) * 16 + r) * FB_FROW + c - sft] = hw;
                fsp[((sft * 2 + 1) * 16 + r) * FB_FROW + c - sft] = lw;
            }
            if (c >= FB_FCOLS - 4 && c - sft + 4 < FB_FROW) {       // the copy's cells past the tile's last column: zeros (never multiplied by a non-zero weight)
                fsp[((sft * 2 + 0) * 16 + r) * FB_FROW + c - sft + 4] = 0u;
                fsp[((sft * 2 + 1) * 16 + r) * FB_FROW + c - sft + 4] = 0u;
            }
        }
    }
    FB_T(2);
    fb_barrier();
    FB_T(3);

    // ---- stage 1: convf1 on the halo cells: 6 row tiles x 4 column tiles = 24 tiles of 32 x 32, three per wave (column tile
    // j1, row tiles (wv >> 2), + 2, + 4).  The MFMAs take the weights as their first operand: D = W1 x flow^T, channels x cells,
    // so that a lane holds 4 x 4 consecutive channels of the cell (lane & 31) -- relu(. + bias), ZERO for a halo cell
    // outside the image (convf2's zero padding), split, eight 8-byte stores into the feature tile: no exchange between lanes.
    const float inv2048 = 1.f / 2048.f;
    {
#pragma unroll 1
        for (int mi = 0; mi < 3; ++mi) {
            const int i = (wv >> 2) + 2 * mi;
            const int m = 32 * i + (lane & 31), mc = m < FB_HCELLS ? m : FB_HCELLS - 1;
            const int ry = mc / FB_HW, rx = mc - ry * FB_HW;
            const int c0 = rx + 4 * (lane >> 5), sft = c0 & 3;
            const unsigned char *arow = lds + FB_OFF_FLOW + (((sft * 2) * 16 + ry) * FB_FROW + (c0 - sft)) * 4;
            // (slots 14, 15 of a filter row carry zero weights: zero operands too, or a non-finite flow one column further would reach this cell as NaN x 0)
            const unsigned padmask = (lane >> 5) ? 0u : 0xffffffffu;
            fb_f32x16 acc1, accx1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc1[r] = 0.f; accx1[r] = 0.f; }
            fb_u32x4 ahw[7], alw[7];
#pragma unroll
            for (int g = 0; g < 7; ++g) {
                ahw[g] = *reinterpret_cast<const fb_u32x4 *>(arow + g * FB_FROW * 4);
                alw[g] = *reinterpret_cast<const fb_u32x4 *>(arow + (16 + g) * FB_FROW * 4);
            }
#pragma unroll
            for (int g = 0; g < 7; ++g) {
                ahw[g][3] &= padmask;
                alw[g][3] &= padmask;
                const fb_f16x8 ah = __builtin_bit_cast(fb_f16x8, ahw[g]), al = __builtin_bit_cast(fb_f16x8, alw[g]);
                acc1 = fb_mfma(w1h[g], ah, acc1);
                accx1 = fb_mfma(w1l[g], ah, accx1);
                accx1 = fb_mfma(w1h[g], al, accx1);
            }
            FB_T(4);
            const int yy = y0 - 1 + ry, xx = x0 - 1 + rx;
            const unsigned keep = (m < FB_HCELLS && yy >= 0 && yy < p.h && xx >= 0 && xx < p.w) ? 0xffffffffu : 0u;
            unsigned char *cellp = lds + m * FB_CELL + (4 * j1) * 32 + (lane >> 5) * 8;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = acc1[4 * b + e] + accx1[4 * b + e] * inv2048 + bias1[b][e];
                    v[e] = (t + __builtin_fabsf(t)) * 0.5f;       // relu that keeps NaN (2 t is exact)
                }
                unsigned h0, l0, h1, l1;
                fb_split_pair(v[0], v[1], k2048, h0, l0);
                fb_split_pair(v[2], v[3], k2048, h1, l1);
                *reinterpret_cast<uint2 *>(cellp + b * 32) = make_uint2(h0 & keep, h1 & keep);
                *reinterpret_cast<uint2 *>(cellp + b * 32 + 16) = make_uint2(l0 & keep, l1 & keep);
            }
            FB_T(5);
        }
    }
    fb_barrier();
    FB_T(6);

    // ---- stage 2: convf2.  Wave (nt, kq): output channels [32 nt, 32 nt + 32) of all four row tiles, channel groups 2 kq, 2 kq + 1 of every tap
    const int nt = wv & 1, kq = wv >> 1;
    const uint4 *__restrict__ w2 = wf + FB_W1_BYTES / 16 + (nt * 4 + kq) * 18 * 128 + lane;
    fb_f32x16 acc[4], accx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; accx[i][r] = 0.f; }
    const unsigned char *abase[4];
    {
        const int r = lane & 31;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            abase[i] = lds + ((2 * i + (r >> 4)) * FB_HW + (r & 15)) * FB_CELL + (lane >> 5) * 32 + kq * 128;
    }
    // Software pipeline, pinned with scheduling barriers (left alone, the compiler hoists every weight load to the top and
    // spills what it fetched): weight fragments PF steps ahead, A fragments one step ahead, both issued in front of the
    // step's 12 MFMAs.
    constexpr int PF = 3;
    uint4 bq[PF][2];
#pragma unroll
    for (int s = 0; s < PF; ++s) { bq[s][0] = w2[s * 128]; bq[s][1] = w2[s * 128 + 64]; }
    fb_f16x8 ah[2][4], al[2][4];
    auto read_a = [&](int s, int set) {
        const int tap = s >> 1, gg = s & 1;
        const int off = ((tap / 3) * FB_HW + tap % 3) * FB_CELL + gg * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ah[set][i] = *reinterpret_cast<const fb_f16x8 *>(abase[i] + off);
            al[set][i] = *reinterpret_cast<const fb_f16x8 *>(abase[i] + off + 16);
        }
    };
    read_a(0, 0);
#pragma unroll
    for (int s = 0; s < 18; ++s) {
        const int set = s & 1;
        const fb_f16x8 bh = __builtin_bit_cast(fb_f16x8, bq[s % PF][0]), bl = __builtin_bit_cast(fb_f16x8, bq[s % PF][1]);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < 18) read_a(s + 1, set ^ 1);
        if (s + PF < 18) { bq[s % PF][0] = w2[(s + PF) * 128]; bq[s % PF][1] = w2[(s + PF) * 128 + 64]; }
        __builtin_amdgcn_sched_barrier(0);
        // (weights first: D = W2 x features^T, channels x cells -- a lane ends up with 4 x 4 consecutive channels of one cell per row tile)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = fb_mfma(bh, ah[set][i], acc[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) accx[i] = fb_mfma(bl, ah[set][i], accx[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) accx[i] = fb_mfma(bh, al[set][i], accx[i]);
        __builtin_amdgcn_sched_barrier(0);
    }
    FB_T(7);
    fb_barrier();           // every wave is done with the feature tile: its space takes the partial sums

    // ---- stage 3: the four K quarters -> LDS [kq][cell][channel], summed in the order kq = 0, 1, 2, 3
    float *red = reinterpret_cast<float *>(lds);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            fb_f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][4 * b + e] + accx[i][4 * b + e] * inv2048;
            *reinterpret_cast<fb_f32x4 *>(red + (kq * 128 + 32 * i + (lane & 31)) * FB_RED_ROW + 32 * nt + 8 * b + 4 * (lane >> 5)) = v;
        }
    FB_T(8);
    fb_barrier();
    FB_T(9);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int item = tid + 512 * it, m = item >> 3, g8 = item & 7;
        const int yy = y0 + (m >> 4), xx = x0 + (m & 15);
        const float *src = red + m * FB_RED_ROW + 8 * g8;
        fb_f32x4 u = *reinterpret_cast<const fb_f32x4 *>(src), v = *reinterpret_cast<const fb_f32x4 *>(src + 4);
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            u += *reinterpret_cast<const fb_f32x4 *>(src + q * 128 * FB_RED_ROW);
            v += *reinterpret_cast<const fb_f32x4 *>(src + q * 128 * FB_RED_ROW + 4);
        }
        const fb_f32x4 bu = *reinterpret_cast<const fb_f32x4 *>(p.b2 + 8 * g8), bv = *reinterpret_cast<const fb_f32x4 *>(p.b2 + 8 * g8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { u[e] = relu_keep_nan(u[e] + bu[e]); v[e] = relu_keep_nan(v[e] + bv[e]); }
        fb_f16x8 hi, lo;
        fb_split8(u, v, k2048, hi, lo);
        if (yy < p.h && xx < p.w) {
            const long long cell = img_base + (long long)yy * p.w + xx;
            uint4 *dst = reinterpret_cast<uint4 *>(reinterpret_cast<char *>(p.out + cell * p.ld_out) + g8 * 32);
            dst[0] = __builtin_bit_cast(uint4, hi);
            dst[1] = __builtin_bit_cast(uint4, lo);
        }
    }
    FB_T(10);
}

#ifdef MFTX_LF_TRACE
extern "C" int mftx_debug_fb_trace(unsigned long long *out) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(fb_trace_buf), sizeof(unsigned long long) * 8 * 16) != hipSuccess) return -1;
    unsigned long long z[8 * 16] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(fb_trace_buf), z, sizeof z) == hipSuccess ? 0 : -1;
}
#endif

// ---- weights: convf1 as [98 = (ky, kx, c)][128] (the direct kernel's form) and convf2 as the GEMM's packed form
// [>= 64 rows][9 taps][128] -> the two fragment streams above, split into fp16 halves
__global__ void pack_flow_branch_kernel(const float *__restrict__ w98, const float *__restrict__ w2pk, uint4 *__restrict__ out) {
    const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);       // one 16-byte piece (8 halves) each
    const int n1 = (int)(FB_W1_BYTES / 16), n2 = (int)(FB_W2_BYTES / 16);
    if (idx >= n1 + n2) return;
    float v[8];
    int part;
    if (idx < n1) {
        const int lane = idx & 63, j = (idx >> 7) & 3, g = idx >> 9;
        part = (idx >> 6) & 1;
        const int n = 32 * j + (lane & 31);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int s = 8 * (lane >> 5) + e;                          // slot in the filter row: kx * 2 + c; 14, 15 are zero
            v[e] = s < 14 ? w98[(g * 14 + s) * 128 + n] : 0.f;
        }
    } else {
        const int k = idx - n1;
        const int lane = k & 63, step = (k >> 7) % 18, kq = ((k >> 7) / 18) & 3, nt = (k >> 7) / 72;
        part = (k >> 6) & 1;
        const int tap = step >> 1, g = 2 * kq + (step & 1);
        const int n = 32 * nt + (lane & 31);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = w2pk[((long long)n * 9 + tap) * 128 + 16 * g + 8 * (lane >> 5) + e];
    }
    unsigned w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned a = split_halves(v[2 * e]), b = split_halves(v[2 * e + 1]);
        w[e] = part ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
    }
    out[idx] = make_uint4(w[0], w[1], w[2], w[3]);
}

int launch_pack_flow_branch(const float *w98, const float *w2pk, void *out, hipStream_t s) {
    const int n = (int)(FB_WBYTES / 16);
    hipLaunchKernelGGL(pack_flow_branch_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w98, w2pk, reinterpret_cast<uint4 *>(out));
    return check_launch("pack_flow_branch");
}

int launch_flow_branch(const float *coords, int P, int h, int w, const void *wf, const float *b1, const float *b2, float *out,
                       int ld_out, float *hx, int ld_hx, hipStream_t s, const float *T, const float *b2h, float *coords_out, float *delta_out) {
    if (T && (!b2h || !coords_out || !delta_out || coords_out == coords || (reinterpret_cast<uintptr_t>(T) & 7) || (reinterpret_cast<uintptr_t>(coords_out) & 7) ||
              (reinterpret_cast<uintptr_t>(delta_out) & 7)))
        return fail(MFTX_E_ARG, "flow_branch: a pending flow-head update needs its bias, a SECOND coordinate buffer and a delta buffer, 8-byte aligned");
    if (!coords || !wf || !b1 || !b2 || !out) return fail(MFTX_E_ARG, "flow_branch: null pointer");
    if (P <= 0 || h <= 0 || w <= 0) return fail(MFTX_E_ARG, "flow_branch: bad sizes");
    if (ld_out < 64 || ld_out % 8 || (reinterpret_cast<uintptr_t>(out) & 31) || (hx && (ld_hx < 384 || ld_hx % 8 || (reinterpret_cast<uintptr_t>(hx) & 31))))
        return fail(MFTX_E_ALIGN, "flow_branch: split-form rows are 32-byte aligned with strides in multiples of 8");
    if (!aligned16(wf) || !aligned16(b2) || (reinterpret_cast<uintptr_t>(coords) & 7)) return fail(MFTX_E_ALIGN, "flow_branch: weights / bias / coordinates misaligned");
    FlowBranchArgs a{coords, wf, b1, b2, out, ld_out, hx, ld_hx, T, b2h, coords_out, delta_out, P, h, w, cdiv(w, FB_TW), cdiv(h, FB_TH)};
    const long long tiles = (long long)P * a.tiles_x * a.tiles_y;
    if (tiles > 0x7fffffffLL) return fail(MFTX_E_ARG, "flow_branch: too many tiles");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(flow_branch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS) != hipSuccess)
            return fail(MFTX_E_STATE, "flow_branch: cannot reserve %d bytes of LDS", FB_LDS);
        attr_set = true;
    }
    ProfScope prof(PC_FLOW_FUSED, s, 2.0 * P * h * w * (98.0 * 128 + 1152.0 * 64));
    hipLaunchKernelGGL(flow_branch_kernel, dim3((unsigned)tiles), dim3(512), FB_LDS, s, a);
    return check_launch("flow_branch");
}

}  // namespace mftx
