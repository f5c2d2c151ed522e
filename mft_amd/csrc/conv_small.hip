// 3x3 convolution with a handful of output channels (N <= 4) over 256 input
// channels: the last layer of the flow head (core/update.py:6-14, 256 -> 2) and
// of the fused occlusion/uncertainty heads (core/update.py:17-75, 256 -> 2 + 1).
//
// An MFMA tile would be >= 87 % padding here (N = 2 of 32 columns), so this is a
// VALU kernel: one wave walks a horizontal strip of cells; lane l owns channels
// 4l..4l+3, keeps its 9 x N x 4 weights in registers for the whole strip, slides
// a 3x3 window of float4 activations along the row (3 new loads per cell instead
// of 9) and reduces the N partial sums across the wave with xor shuffles.
// Reads are 1 KiB-coalesced per (row, column); traffic is 3 (len + 2) / len x the input map.
#include "common.h"
#include "profile.h"
#include <cstdlib>

namespace mftx {


template <int N>
__global__ __launch_bounds__(256) void conv3x3_small_kernel(const float *__restrict__ x, int ldx,
                                                            const float *__restrict__ wpk,   // [>=N][9][256]
                                                            const float *__restrict__ bias, float *__restrict__ out,
                                                            int ldo, int P, int h, int w, int strip_len, int strips_per_row,
                                                            float *__restrict__ accum, int ld_accum) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);          // global wave id
    const int total = P * h * strips_per_row;
    if (gw >= total) return;
    const int strip = gw % strips_per_row;
    const int rowid = gw / strips_per_row;                        // img*h + y
    const int y = rowid % h;
    const long long img_base = (long long)(rowid / h) * h * w;
    const int x0 = strip * strip_len;
    const int x1 = min(x0 + strip_len, w);

    float4 wt[9][N];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int n = 0; n < N; ++n)
            wt[t][n] = *reinterpret_cast<const float4 *>(wpk + ((long long)n * 9 + t) * 256 + lane * 4);
    // lane t * N + n keeps output channel n of the strip's cell t: one store pass after the loop (a store
    // inside it would sit between the loads of consecutive cells and drain vmcnt every time)
    const float bb = bias != nullptr ? bias[lane % N] : 0.f;
    float keep = 0.f;
    const bool mine = lane < (x1 - x0) * N;
    const long long my_cell = img_base + (long long)y * w + x0 + lane / N;
    // the value to accumulate into is fetched now, so that its latency is gone by the end of the strip
    const float old = (mine && accum != nullptr) ? accum[my_cell * ld_accum + lane % N] : 0.f;

    struct Col { float4 r0, r1, r2; };
    const bool y0ok = y - 1 >= 0, y2ok = y + 1 < h;
    auto ld = [&](bool ok, int yy, int xx) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v = *reinterpret_cast<const float4 *>(x + (img_base + (long long)yy * w + xx) * ldx + lane * 4);
        return v;
    };
    auto load_col = [&](int xx) {
        const bool xok = xx >= 0 && xx < w;
        Col c;
        c.r0 = ld(xok && y0ok, y - 1, xx);
        c.r1 = ld(xok, y, xx);
        c.r2 = ld(xok && y2ok, y + 1, xx);
        return c;
    };
    auto dot4 = [](const float4 &a, const float4 &b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; };
    Col c0 = load_col(x0 - 1), c1 = load_col(x0);      // columns x-1, x; x+1 is loaded per step
    for (int xc = x0; xc < x1; ++xc) {
        const Col c2 = load_col(xc + 1);
        float acc[N];
#pragma unroll
        for (int n = 0; n < N; ++n) {
            float a = dot4(c0.r0, wt[0][n]) + dot4(c1.r0, wt[1][n]) + dot4(c2.r0, wt[2][n]);
            a += dot4(c0.r1, wt[3][n]) + dot4(c1.r1, wt[4][n]) + dot4(c2.r1, wt[5][n]);
            a += dot4(c0.r2, wt[6][n]) + dot4(c1.r2, wt[7][n]) + dot4(c2.r2, wt[8][n]);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off);
            acc[n] = a;
        }
#pragma unroll
        for (int n = 0; n < N; ++n)
            if (lane == (xc - x0) * N + n) keep = acc[n] + bb;
        c0 = c1;
        c1 = c2;
    }
    if (mine) {
        out[my_cell * ldo + lane % N] = keep;
        if (accum != nullptr) accum[my_cell * ld_accum + lane % N] = old + keep;   // coords1 += delta_flow (core/raft.py:184)
    }
}

bool conv_small_applicable(const mftx_conv_desc &d) {
    return d.N <= 4 && d.kh == 3 && d.kw == 3 && d.c0 == 256 && d.c1 == 0 && d.act == 0 && d.out_scale == 1.0f &&
           d.lda0 % 4 == 0;
}

// accum (optional, [cells][ld_accum]): the N outputs are also added to it -- the flow head's last layer
// updates coords1 in the same pass instead of a separate add kernel.
int launch_conv_small(const mftx_conv_desc &d, hipStream_t s, float *accum, int ld_accum) {
    // Cells per wave (measured, tools/bench_small.py): 16 at 7 pairs (20.5 us; 3 -> 24.1 us: the halo
    // columns of short strips are re-read), 8 when one or two pairs leave the chip short of waves
    // (13.7 -> 9.9 us at one pair).
    static const int forced = [] { const char *e = getenv("MFTX_SMALL_STRIP"); return e ? atoi(e) : 0; }();
    const long long cells = (long long)d.P * d.h * d.w;
    const int strip_len = forced > 0 ? forced : (cells >= 16384 ? 16 : 8);
    const int strips = cdiv(d.w, strip_len);
    const int waves = d.P * d.h * strips;
    dim3 grid(cdiv(waves, 4));
    ProfScope prof(PC_CONV_SMALL, s, 2.0 * d.P * d.h * d.w * d.N * 9.0 * 256.0);
#define SN_LAUNCH(NN)                                                                                              \
    hipLaunchKernelGGL(conv3x3_small_kernel<NN>, grid, dim3(256), 0, s, d.a0, d.lda0, d.wpk, d.bias, d.out, d.ldo, \
                       d.P, d.h, d.w, strip_len, strips, accum, ld_accum)
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
