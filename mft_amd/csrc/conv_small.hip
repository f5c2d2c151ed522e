// 3x3 convolution with a handful of output channels (N <= 4) over 256 input
// channels: the last layer of the flow head (core/update.py:6-14, 256 -> 2) and
// of the fused occlusion/uncertainty heads (core/update.py:17-75, 256 -> 2 + 1).
//
// An MFMA tile would be >= 87 % padding here (N = 2 of 32 columns), so this is a
// VALU kernel (see the kernel's comment).  Reads are 1 KiB-coalesced per (row, column);
// traffic is 3 (len + 2) / len x the input map, nearly all of it L2 hits (the
// producing GEMM has just written the 29 MB map).
#include "common.h"
#include "profile.h"
#include <cstdlib>

namespace mftx {


typedef float f32x4s __attribute__((ext_vector_type(4)));

// One wave per strip of SMALL_STRIP cells of one row.  Lane l owns input channels 4l..4l+3.
//   * every load of the strip -- 3 rows x (strip + 2) columns of float4, raw buffer loads whose offset is
//     out of range on the zero-padding border (the hardware returns zeros, no branches) -- is issued before
//     the first use: one memory round trip per strip (the first version walked the strip column by column and
//     paid the L2 latency 16 times; 48 us for a 29 MB read);
//   * the 9 x N x 4 weights of a lane sit in LDS (shared by the block's 4 waves), read as float4 when used;
//   * each lane accumulates its 4-channel partial sums for all strip x N outputs, and ONE transposing butterfly
//     reduces them across the wave: at every step a lane keeps half of its values and trades the other half with
//     its partner (lane ^ 32, ^ 16, ...), so 8 N values take 8 N - 1 + (a few) shuffles instead of 6 per
//     value, and lane 4 v ends up with output v = cell * N + n.  Fixed order: results do not depend on batch or grid.
// strip length: 8 cells for N <= 2 (the flow head, 12 launches per refinement), 4 for N = 3, 4 (the OU heads, one launch):
// keeps 3 x (strip + 2) float4 columns + strip x N partial sums within the register budget
constexpr int small_strip(int n) { return n <= 2 ? 8 : 4; }

template <int N>
__global__ __launch_bounds__(256, 2) void conv3x3_small_kernel(const float *__restrict__ x, int ldx, unsigned x_bytes,
                                                            const float *__restrict__ wpk,   // [>=N][9][256]
                                                            const float *__restrict__ bias, float *__restrict__ out,
                                                            int ldo, int P, int h, int w, int strips_per_row,
                                                            float *__restrict__ accum, int ld_accum) {
    constexpr int S = small_strip(N);
    constexpr int NP = N <= 2 ? 2 : 4;                  // values per cell in the reduction (power of two)
    constexpr int V = S * NP;
    __shared__ __attribute__((aligned(16))) float wsm[9 * N * 256];
    const int lane = threadIdx.x & 63;
    // A block = 4 consecutive rows of one strip (its waves share 2 of their 3 input rows), and workgroup b runs
    // on XCD b % 8 (observed; only speed depends on it): each XCD gets a contiguous band of row groups, strips
    // innermost, so the halo rows / columns a wave re-reads were fetched by a neighbour on the SAME XCD and hit
    // its L2 -- the input map (written by another XCD's GEMM tiles) crosses the fabric ~once instead of 3.75 x.
    const int rows = P * h, groups = (rows + 3) / 4;
    const int nb = groups * strips_per_row, per_xcd = (nb + 7) / 8;
    const int vb = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || vb >= nb) return;            // (uniform per block)
    const int strip = vb % strips_per_row;
    const int rowid_raw = (vb / strips_per_row) * 4 + (threadIdx.x >> 6); // img*h + y
    const bool live = rowid_raw < rows;                                   // a wave past the last row still helps fill the weights
    const int rowid = live ? rowid_raw : rows - 1;
    const int y = rowid % h;
    const int img_base = (rowid / h) * h * w;
    const int x0 = strip * S;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, x_bytes, 0x00020000);

    f32x4s col[3][S + 2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int yy = y + r - 1;
        const bool yok = yy >= 0 && yy < h;
#pragma unroll
        for (int c = 0; c < S + 2; ++c) {
            const int xx = x0 + c - 1;
            const bool ok = yok && xx >= 0 && xx < w;
            const unsigned off = ok ? ((unsigned)(img_base + yy * w + xx) * (unsigned)ldx + (unsigned)lane * 4u) * 4u : 0x80000000u;
            col[r][c] = __builtin_bit_cast(f32x4s, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        }
    }
    // the block's weights go to LDS while the input loads are in flight (both latencies overlap)
    for (int i = threadIdx.x; i < 9 * N * 64; i += 256)
        reinterpret_cast<f32x4s *>(wsm)[i] = reinterpret_cast<const f32x4s *>(wpk)[i];     // [n][tap][256] as packed
    __syncthreads();
    if (!live) return;
    // the value to accumulate into is fetched now, so that its latency is gone by the end of the strip
    const int v_mine = lane >> 2;                                  // output this lane ends up with (if lane % 4 == 0)
    const int t_mine = v_mine / NP, n_mine = v_mine % NP;
    const bool mine = (lane & 3) == 0 && n_mine < N && x0 + t_mine < w;
    const long long my_cell = (long long)img_base + (long long)y * w + x0 + t_mine;
    const float old = (mine && accum != nullptr) ? accum[my_cell * ld_accum + n_mine] : 0.f;
    const float bb = (mine && bias != nullptr) ? bias[n_mine] : 0.f;

    float v[V];
#pragma unroll
    for (int i = 0; i < V; ++i) v[i] = 0.f;
    auto dot4 = [](const f32x4s &a, const f32x4s &b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; };
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int n = 0; n < N; ++n) {
                const f32x4s wv = *reinterpret_cast<const f32x4s *>(wsm + ((n * 9 + r * 3 + kx) * 256 + lane * 4));
#pragma unroll
                for (int t = 0; t < S; ++t) v[t * NP + n] += dot4(col[r][t + kx], wv);
            }
    // transposing butterfly: V values -> 1 per lane
    int cnt = V;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const bool upper = (lane & off) != 0;
        if (cnt > 1) {
            const int half = cnt / 2;
#pragma unroll
            for (int i = 0; i < half; ++i) {
                const float send = upper ? v[i] : v[i + half];
                const float keep = upper ? v[i + half] : v[i];
                v[i] = keep + __shfl_xor(send, off);
            }
            cnt = half;
        } else {
            v[0] += __shfl_xor(v[0], off);
        }
    }
    if (mine) {
        const float keep = v[0] + bb;
        out[my_cell * ldo + n_mine] = keep;
        if (accum != nullptr) accum[my_cell * ld_accum + n_mine] = old + keep;   // coords1 += delta_flow (core/raft.py:184)
    }
}

bool conv_small_applicable(const mftx_conv_desc &d) {
    return d.N <= 4 && d.kh == 3 && d.kw == 3 && d.c0 == 256 && d.c1 == 0 && d.act == 0 && d.out_scale == 1.0f &&
           d.lda0 % 4 == 0;
}

// accum (optional, [cells][ld_accum]): the N outputs are also added to it -- the flow head's last layer
// updates coords1 in the same pass instead of a separate add kernel.
int launch_conv_small(const mftx_conv_desc &d, hipStream_t s, float *accum, int ld_accum) {
    const long long cells = (long long)d.P * d.h * d.w;
    if (cells * d.lda0 * 4 > 0x7fffffffLL) return fail(MFTX_E_ARG, "conv3x3_small: activation operand exceeds 2 GiB");
    const int strips = cdiv(d.w, small_strip(d.N));
    const int nb = cdiv(d.P * d.h, 4) * strips;
    const unsigned x_bytes = (unsigned)(((cells - 1) * d.lda0 + 256) * 4);
    dim3 grid(8 * cdiv(nb, 8));
    ProfScope prof(PC_CONV_SMALL, s, 2.0 * d.P * d.h * d.w * d.N * 9.0 * 256.0);
#define SN_LAUNCH(NN)                                                                                              \
    hipLaunchKernelGGL(conv3x3_small_kernel<NN>, grid, dim3(256), 0, s, d.a0, d.lda0, x_bytes, d.wpk, d.bias, d.out, \
                       d.ldo, d.P, d.h, d.w, strips, accum, ld_accum)
    switch (d.N) {
        case 1: SN_LAUNCH(1); break;
        case 2: SN_LAUNCH(2); break;
        case 3: SN_LAUNCH(3); break;
        default: SN_LAUNCH(4); break;
    }
#undef SN_LAUNCH
    return check_launch("conv3x3_small");
}

}  // namespace mftx
