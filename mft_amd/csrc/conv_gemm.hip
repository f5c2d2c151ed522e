// fp32-MFMA implicit-GEMM convolution / NT-GEMM for gfx950 (MI355X).
//
//   out[m][n] = epilogue( sum_{tap,c} A[cell(m)+off(tap)][c] * W[n][tap][c] + bias[n] )
//
// A is pixel-major (NHWC) so the reduction axis is contiguous for both
// operands; the same kernel serves
//   * every conv of the RAFT update block and the OU heads (core/update.py),
//   * the all-pairs correlation volume f1 . f2^T / sqrt(C) (core/corr.py:53-69),
//     with kh = kw = 1 and "weights" = the second feature map.
//
// Tiling: 256 threads = 4 waves; block tile BM x BN, K step 32; each wave owns
// (BM/WM) x (BN/WN) as 32x32 v_mfma_f32_32x32x2_f32 tiles (exact fp32, 157 TF
// peak).  Operands are staged global -> VGPR -> LDS (zero-fill for the conv
// halo and ragged channel counts), double-buffered, rows padded to 36 floats so
// that the ds_read_b128 fragment reads are bank-conflict free.  Each lane reads
// 4 consecutive k of its row; lanes 0-31 take k 0..3 and lanes 32-63 k 4..7 of an
// 8-wide group, which is a legal permutation of the reduction order as long as
// A and B use the same one.
#include "common.h"
#include "profile.h"

namespace mftx {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int LDK = BK + 4;  // padded LDS row (floats)

struct ConvArgs {
    const float *a0, *a1;
    int lda0, lda1, c0, c1;
    const float *w;
    const float *bias;
    float *out;
    int ldo;
    int M, N, h, wd, kh, kw, cin_pad;
    int w_rows;               // valid rows of the W operand
    int act;
    float out_scale;
    long long a_bstride, w_bstride, o_bstride;  // per blockIdx.z (correlation batch)
    int gru_mode;
    float *hx; int ld_hx; float *z; float *rh;
};

__device__ __forceinline__ float act_fn(float v, int act) {
    switch (act) {
        case 1: return fmaxf(v, 0.f);
        case 2: return 1.f / (1.f + expf(-v));
        case 3: return tanhf(v);
        default: return v;
    }
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_gemm_kernel(ConvArgs p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int RA = BM / 32, RB = BN / 32;
    static_assert(WM * WN == 4, "4 waves");
    static_assert(TM >= 1 && TN >= 1, "tile");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;                       // [2][BM][LDK]
    float *Bs = smem + 2 * BM * LDK;        // [2][BN][LDK]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const float *a0 = p.a0 + blockIdx.z * p.a_bstride;
    const float *a1 = p.a1;
    const float *wgt = p.w + blockIdx.z * p.w_bstride;
    float *out = p.out + blockIdx.z * p.o_bstride;

    // ---- per-thread staging coordinates
    const int col4 = (tid & 7) * 4;
    const int srow = tid >> 3;  // 0..31
    int ay[RA], ax[RA], am[RA];
    const int hw = p.h * p.wd;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
        int m = m0 + srow + 32 * i;
        if (m < p.M) {
            int rem = m % hw;
            ay[i] = rem / p.wd;
            ax[i] = rem - ay[i] * p.wd;
            am[i] = m;
        } else {
            ay[i] = -100000; ax[i] = -100000; am[i] = 0;
        }
    }
    const int taps = p.kh * p.kw;
    const int cpt = p.cin_pad / BK;          // chunks per tap
    const int T = taps * cpt;
    const long long ktot = (long long)taps * p.cin_pad;
    const int py = p.kh / 2, px = p.kw / 2;
    const int ctot = p.c0 + p.c1;

    f32x4 ra[RA], rb[RB];

    auto load_tiles = [&](int it) {
        const int tap = it / cpt;
        const int cc = it - tap * cpt;
        const int dy = tap / p.kw - py;
        const int dx = tap % p.kw - px;
        const int c = cc * BK + col4;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const int yy = ay[i] + dy, xx = ax[i] + dx;
            const bool ok = (yy >= 0) & (yy < p.h) & (xx >= 0) & (xx < p.wd) & (c < ctot);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) {
                const long long src = (long long)am[i] + dy * p.wd + dx;
                const float *ptr = (c < p.c0) ? (a0 + src * p.lda0 + c) : (a1 + src * p.lda1 + (c - p.c0));
                v = *reinterpret_cast<const f32x4 *>(ptr);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int n = n0 + srow + 32 * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (n < p.w_rows)
                v = *reinterpret_cast<const f32x4 *>(wgt + (long long)n * ktot + (long long)it * BK + col4);
            rb[i] = v;
        }
    };
    auto store_tiles = [&](int buf) {
        float *as = As + buf * BM * LDK;
        float *bs = Bs + buf * BN * LDK;
#pragma unroll
        for (int i = 0; i < RA; ++i)
            *reinterpret_cast<f32x4 *>(as + (srow + 32 * i) * LDK + col4) = ra[i];
#pragma unroll
        for (int i = 0; i < RB; ++i)
            *reinterpret_cast<f32x4 *>(bs + (srow + 32 * i) * LDK + col4) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tiles(0);
    store_tiles(0);
    __syncthreads();

    const int a_row0 = wm * TM * 32 + (lane & 31);
    const int b_row0 = wn * TN * 32 + (lane & 31);
    const int khalf = (lane >> 5) * 4;

    for (int it = 0; it < T; ++it) {
        const int buf = it & 1;
        if (it + 1 < T) load_tiles(it + 1);
        const float *as = As + buf * BM * LDK;
        const float *bs = Bs + buf * BN * LDK;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            f32x4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *reinterpret_cast<const f32x4 *>(as + (a_row0 + 32 * i) * LDK + kk * 8 + khalf);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[j] = *reinterpret_cast<const f32x4 *>(bs + (b_row0 + 32 * j) * LDK + kk * 8 + khalf);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
        }
        if (it + 1 < T) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int col_l = lane & 31;
    const int row_h = 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * TN * 32 + j * 32 + col_l;
        const bool n_ok = n < p.N;
        const float bias = (p.bias != nullptr && n_ok) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + row_h;
                if (!n_ok || m >= p.M) continue;
                float v = act_fn(acc[i][j][r] + bias, p.act) * p.out_scale;
                if (p.gru_mode == 0) {
                    out[(long long)m * p.ldo + n] = v;
                } else if (p.gru_mode == 1) {     // [z | r] gates; r is folded into r*h
                    if (n < 128) p.z[(long long)m * 128 + n] = v;
                    else p.rh[(long long)m * 128 + (n - 128)] = v * p.hx[(long long)m * p.ld_hx + (n - 128)];
                } else {                          // candidate q, h <- (1-z) h + z q
                    const float zz = p.z[(long long)m * 128 + n];
                    float *hp = p.hx + (long long)m * p.ld_hx + n;
                    *hp = (1.f - zz) * (*hp) + zz * v;
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
static int launch_cfg(const ConvArgs &a, int batch, hipStream_t s, ProfCat cat = PC_CONV_GEMM) {
    constexpr size_t lds = 2ull * (BM + BN) * LDK * sizeof(float);
    static bool attr_set = false;
    auto kern = conv_gemm_kernel<BM, BN, WM, WN>;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    dim3 grid(cdiv(a.M, BM), cdiv(a.N, BN), batch);
    // algorithmic flops: real (unpadded) reduction length
    ProfScope prof(cat, s, 2.0 * a.M * a.N * (double)(a.kh * a.kw) * (a.c0 + a.c1) * batch);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    return check_launch("conv_gemm");
}

static int dispatch(const ConvArgs &a, int batch, hipStream_t s) {
    // Tile choice: the largest tile that still yields >= ~1.5 workgroups per CU
    // (256 CUs); N <= 32 (flow / OU output heads) gets a 128x32 tile.
    const long long M = a.M, N = a.N;
    if (N <= 32) return launch_cfg<128, 32, 4, 1>(a, batch, s);
    auto blocks = [&](int bm, int bn) { return (long long)cdiv((int)M, bm) * cdiv((int)N, bn) * batch; };
    if (N % 128 == 0 && blocks(128, 128) >= 384) return launch_cfg<128, 128, 2, 2>(a, batch, s);
    if (blocks(128, 64) >= 384) return launch_cfg<128, 64, 2, 2>(a, batch, s);
    return launch_cfg<64, 64, 2, 2>(a, batch, s);
}

static int validate(const mftx_conv_desc &d) {
    if (!d.a0 || !d.wpk || !d.out) return fail(MFTX_E_ARG, "conv2d: null pointer");
    if (d.P <= 0 || d.h <= 0 || d.w <= 0 || d.N <= 0 || d.c0 <= 0 || d.c1 < 0)
        return fail(MFTX_E_ARG, "conv2d: bad sizes");
    if (d.kh < 1 || d.kw < 1 || !(d.kh & 1) || !(d.kw & 1)) return fail(MFTX_E_ARG, "conv2d: odd kernels only");
    if (d.c1 > 0 && (!d.a1 || d.c0 % BK)) return fail(MFTX_E_ARG, "conv2d: segment 0 must be a multiple of 32 channels");
    if ((d.c0 + d.c1) % 4 || d.c0 % 4) return fail(MFTX_E_ARG, "conv2d: channel counts must be multiples of 4");
    if (d.lda0 % 4 || (d.c1 > 0 && d.lda1 % 4) || !aligned16(d.a0) || (d.c1 > 0 && !aligned16(d.a1)) || !aligned16(d.wpk))
        return fail(MFTX_E_ALIGN, "conv2d: operands must be 16-byte aligned");
    if (d.act < 0 || d.act > 3) return fail(MFTX_E_ARG, "conv2d: bad activation");
    if ((long long)d.P * d.h * d.w > 0x7fffffffLL) return fail(MFTX_E_ARG, "conv2d: too many cells");
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
    a.gru_mode = 0;
    return a;
}

int launch_conv(const mftx_conv_desc &d, hipStream_t s) {
    if (int e = validate(d)) return e;
    return dispatch(to_args(d), 1, s);
}

int launch_conv_gru(const mftx_conv_desc &d, const GruEpilogue &g, hipStream_t s) {
    if (int e = validate(d)) return e;
    ConvArgs a = to_args(d);
    a.gru_mode = g.mode; a.hx = g.hx; a.ld_hx = g.ld_hx; a.z = g.z; a.rh = g.rh;
    if ((g.mode == 1 && d.N != 256) || (g.mode == 2 && d.N != 128)) return fail(MFTX_E_ARG, "gru epilogue: bad N");
    return dispatch(a, 1, s);
}

// lvl0[p][i][j] = <f1[p][i][:], f2[p][j][:]> / sqrt(C)
int launch_corr_volume(const float *f1, const float *f2, int P, int C, int N, float *lvl0, hipStream_t s) {
    ConvArgs a{};
    a.a0 = f1; a.lda0 = C; a.c0 = C; a.c1 = 0; a.a1 = nullptr; a.lda1 = 0;
    a.w = f2; a.bias = nullptr; a.out = lvl0; a.ldo = N;
    a.M = N; a.N = N; a.h = 1; a.wd = N; a.kh = 1; a.kw = 1; a.cin_pad = C;
    a.w_rows = N;
    a.act = 0; a.out_scale = 1.0f / sqrtf((float)C);
    a.a_bstride = (long long)N * C; a.w_bstride = (long long)N * C; a.o_bstride = (long long)N * N;
    a.gru_mode = 0;
    if (N >= 1024) return launch_cfg<128, 128, 2, 2>(a, P, s, PC_CORR_VOLUME);
    return launch_cfg<64, 64, 2, 2>(a, P, s, PC_CORR_VOLUME);
}

}  // namespace mftx
