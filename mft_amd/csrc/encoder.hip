// Feature / context encoders (BasicEncoder, core/extractor.py:118-195) on the
// fp32-MFMA implicit-GEMM kernel, pixel-major end to end:
//
//   prep      uint8 BGR frame -> RGB, 2x/255-1, replicate pad to /8 (InputPadder 'sintel',
//             core/utils/utils.py:9-19), 4th channel and 3 zero columns each side
//   stem      7x7 stride-2 conv 3->64 as a 7-tap GEMM over 28-float windows (7 pixels x RGBX)
//   3 stages  of 2 residual blocks (64 s1, 96 s2, 128 s2): 3x3 convs, 1x1 stride-2 shortcut
//   head      1x1 conv 128->256 (fnet) or 128->128 tanh | 128->128 relu (cnet, core/raft.py:146-149)
//
// fnet uses InstanceNorm2d (no affine, eps 1e-5): one statistics pass (fp64 sums, fixed
// reduction order -> deterministic) + one normalise/ReLU(/residual) pass per conv.  cnet
// uses eval-mode BatchNorm2d, folded into the conv weights and biases at load, so every
// layer is a single GEMM launch with a fused ReLU or residual epilogue.
#include "common.h"
#include "profile.h"
#include "graph_cache.h"
#include <new>

namespace mftx {

// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------

// img: uint8 [H0][W0][3] BGR  ->  out: fp32 [Hp][Wp + 6][4] (RGB0), Hp x Wp = padded size
__global__ void enc_prep_kernel(const uint8_t *__restrict__ img, int H0, int W0, int pl, int pt, int Hp, int Wp,
                                float *__restrict__ out) {
    const int xo = blockIdx.x * blockDim.x + threadIdx.x;     // 0 .. Wp+5
    const int y = blockIdx.y;
    if (xo >= Wp + 6) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int x = xo - 3;
    if (x >= 0 && x < Wp) {
        const int sy = min(max(y - pt, 0), H0 - 1), sx = min(max(x - pl, 0), W0 - 1);   // replicate padding
        const uint8_t *p = img + ((long long)sy * W0 + sx) * 3;
        v.x = 2.f * ((float)p[2] / 255.0f) - 1.0f;     // R   (core/raft.py:122-124: 2*(x/255) - 1)
        v.y = 2.f * ((float)p[1] / 255.0f) - 1.0f;     // G
        v.z = 2.f * ((float)p[0] / 255.0f) - 1.0f;     // B
    }
    reinterpret_cast<float4 *>(out)[(long long)y * (Wp + 6) + xo] = v;
}

// per-channel partial sums over a slab of rows: x [rows][C] -> part [slabs][C][2] (sum, sum of squares; fp64).
// A thread owns 4 consecutive channels (one 16-byte load per row) and every (256 / (C/4))-th row of the
// slab; the row groups of a block are then summed in a fixed order.
constexpr int IN_SLABS = 256;   // row slabs per map: enough workgroups to stream at HBM rate
__global__ __launch_bounds__(256) void instnorm_partial_kernel(const float *__restrict__ x, int rows, int C,
                                                               double *__restrict__ part) {
    __shared__ double sh[256][8];
    const int n4 = C >> 2;                             // float4 columns (C <= 256 -> n4 <= 64)
    const int rpp = 256 / n4;                          // rows per pass
    const int g = threadIdx.x / n4, c4 = threadIdx.x - g * n4;
    const int slab = blockIdx.x;
    const int r0 = (int)((long long)rows * slab / IN_SLABS), r1 = (int)((long long)rows * (slab + 1) / IN_SLABS);
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (g < rpp) {
        const float4 *x4 = reinterpret_cast<const float4 *>(x);
        int r = r0 + g;
        for (; r + 3 * rpp < r1; r += 4 * rpp) {       // four independent loads in flight
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = x4[(long long)(r + u * rpp) * n4 + c4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double d0 = v[u].x, d1 = v[u].y, d2 = v[u].z, d3 = v[u].w;
                acc[0] += d0; acc[1] += d0 * d0; acc[2] += d1; acc[3] += d1 * d1;
                acc[4] += d2; acc[5] += d2 * d2; acc[6] += d3; acc[7] += d3 * d3;
            }
        }
        for (; r < r1; r += rpp) {
            const float4 v = x4[(long long)r * n4 + c4];
            const double d0 = v.x, d1 = v.y, d2 = v.z, d3 = v.w;
            acc[0] += d0; acc[1] += d0 * d0; acc[2] += d1; acc[3] += d1 * d1;
            acc[4] += d2; acc[5] += d2 * d2; acc[6] += d3; acc[7] += d3 * d3;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) sh[threadIdx.x][k] = acc[k];
    __syncthreads();
    if (g == 0) {
        for (int k = 1; k < rpp; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += sh[threadIdx.x + k * n4][j];
        double *o = part + ((long long)slab * C + c4 * 4) * 2;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = acc[j];
    }
}

// partial sums -> per-channel (mean, rstd), in a fixed order (deterministic): a block owns 16 channels,
// 16 threads per channel each add every 16th slab, then one thread per channel adds the 16 sums
__global__ __launch_bounds__(256) void instnorm_finalize_kernel(const double *__restrict__ part, int rows, int C,
                                                                float *__restrict__ stat) {
    __shared__ double sh[16][16][2];
    const int cl = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s = 0.0, q = 0.0;
    if (c < C) {
#pragma unroll
        for (int k = 0; k < IN_SLABS / 16; ++k) {
            const double2 v = *reinterpret_cast<const double2 *>(part + ((long long)(g + 16 * k) * C + c) * 2);
            s += v.x;
            q += v.y;
        }
    }
    sh[g][cl][0] = s;
    sh[g][cl][1] = q;
    __syncthreads();
    if (g == 0 && c < C) {
        for (int k = 1; k < 16; ++k) { s += sh[k][cl][0]; q += sh[k][cl][1]; }
        const double m = s / rows;
        const double var = q / rows - m * m;           // biased variance, as nn.InstanceNorm2d
        stat[2 * c] = (float)m;
        stat[2 * c + 1] = (float)(1.0 / sqrt((var > 0 ? var : 0.0) + 1e-5));
    }
}

// y = relu((x - mean) * rstd)            [mode 0]
// y = relu(res + relu((x - mean)*rstd))  [mode 1: residual block tail]
// y = (x - mean) * rstd                  [mode 2: shortcut branch, no activation]
// split != 0 (split arithmetic, round 4): y is written IN SPLIT FORM (common.h) -- the convolution that reads it then finds its A
// operand split instead of splitting it in registers for every output tile -- and `res`, an earlier output of this kernel, is read
// from it.  A thread owns 8 consecutive channels: the 32 bytes of the group it reads are the 32 bytes it writes (in place).
__global__ __launch_bounds__(256) void instnorm_apply_kernel(float *__restrict__ x, int rows, int C,
                                                             const float *__restrict__ stat,
                                                             const float *__restrict__ res, int mode, int split) {
    __shared__ float mean_s[256], rstd_s[256];
    for (int c = threadIdx.x; c < C; c += blockDim.x) { mean_s[c] = stat[2 * c]; rstd_s[c] = stat[2 * c + 1]; }
    __syncthreads();
    const long long n8 = (long long)rows * C / 8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const long long e = i * 8, r = e / C;
        const int c = (int)(e - r * C);
        const float4 va = reinterpret_cast<const float4 *>(x)[2 * i], vb = reinterpret_cast<const float4 *>(x)[2 * i + 1];
        float o[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
        float rr[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (mode == 1) {
            float4 ra, rb;
            if (split) { ra = load_split4(res + r * C, c); rb = load_split4(res + r * C, c + 4); }
            else { ra = reinterpret_cast<const float4 *>(res)[2 * i]; rb = reinterpret_cast<const float4 *>(res)[2 * i + 1]; }
            rr[0] = ra.x; rr[1] = ra.y; rr[2] = ra.z; rr[3] = ra.w; rr[4] = rb.x; rr[5] = rb.y; rr[6] = rb.z; rr[7] = rb.w;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float t = (o[k] - mean_s[c + k]) * rstd_s[c + k];
            if (mode != 2) t = relu_keep_nan(t);
            if (mode == 1) t = relu_keep_nan(rr[k] + t);
            o[k] = t;
        }
        if (split) {
            store_split4v(x + r * C, c, make_float4(o[0], o[1], o[2], o[3]));
            store_split4v(x + r * C, c + 4, make_float4(o[4], o[5], o[6], o[7]));
        } else {
            reinterpret_cast<float4 *>(x)[2 * i] = make_float4(o[0], o[1], o[2], o[3]);
            reinterpret_cast<float4 *>(x)[2 * i + 1] = make_float4(o[4], o[5], o[6], o[7]);
        }
    }
}

// ---------------------------------------------------------------------------
// engine
// ---------------------------------------------------------------------------
// weight slots per encoder: 16 convs x (weight, bias); the cnet head is two convs
enum EncConv { EC_STEM, EC_L1B0C1, EC_L1B0C2, EC_L1B1C1, EC_L1B1C2, EC_L2B0C1, EC_L2B0C2, EC_L2B0DS, EC_L2B1C1,
               EC_L2B1C2, EC_L3B0C1, EC_L3B0C2, EC_L3B0DS, EC_L3B1C1, EC_L3B1C2, EC_HEAD, EC_HEAD2, EC_COUNT };

struct EncWs {
    float *img;           // [Hp][Wp+6][4]
    float *a, *b, *c;     // activation ping-pong, each max(stage maps)
    double *part;         // instance-norm partial sums
    float *stat;          // per-channel (mean, rstd)
    size_t bytes;
};

static EncWs enc_carve(void *base, int Hp, int Wp) {
    EncWs ws{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *p = base ? static_cast<char *>(base) + off : nullptr;
        off += (bytes + 255) & ~size_t(255);
        return p;
    };
    const size_t big = (size_t)(Hp / 2) * (Wp / 2) * 64 * sizeof(float);   // stage-1 map, the largest
    ws.img = reinterpret_cast<float *>(take((size_t)Hp * (Wp + 6) * 4 * sizeof(float)));
    ws.a = reinterpret_cast<float *>(take(big));
    ws.b = reinterpret_cast<float *>(take(big));
    ws.c = reinterpret_cast<float *>(take(big));
    ws.part = reinterpret_cast<double *>(take((size_t)IN_SLABS * 256 * 2 * sizeof(double)));
    ws.stat = reinterpret_cast<float *>(take(256 * 2 * sizeof(float)));
    ws.bytes = off;
    return ws;
}

}  // namespace mftx

using namespace mftx;

struct mftx_encoder {
    uint32_t magic;
    int instance_norm;          // 1: fnet (instance norm), 0: cnet (batch norm folded into the weights)
    const float *w[EC_COUNT], *b[EC_COUNT];
    const float *wg[EC_COUNT];  // what the conv GEMMs stream: w, or its split form (arith = MFTX_ARITH_SPLIT)
    int arith;
    int use_graph;              // replay the layers between the pre-processing kernel and the head as a hipGraph (graph_cache.h)
    GraphCache *graphs;
};
static constexpr uint32_t ENC_MAGIC = 0x454e4358;

extern "C" int mftx_encoder_create(const float *const *weights, int n_weights, int instance_norm, mftx_encoder **out) {
    if (!weights || !out) return fail(MFTX_E_ARG, "encoder_create: null pointer");
    const int n_conv = instance_norm ? EC_COUNT - 1 : EC_COUNT;
    if (n_weights != 2 * n_conv) return fail(MFTX_E_ARG, "encoder_create: expected %d tensors, got %d", 2 * n_conv, n_weights);
    for (int i = 0; i < n_weights; ++i)
        if (!weights[i] || !aligned16(weights[i])) return fail(MFTX_E_ALIGN, "encoder_create: tensor %d null or unaligned", i);
    mftx_encoder *e = new (std::nothrow) mftx_encoder;
    if (!e) return fail(MFTX_E_ARG, "encoder_create: out of host memory");
    e->magic = ENC_MAGIC;
    e->instance_norm = instance_norm ? 1 : 0;
    for (int i = 0; i < EC_COUNT; ++i) { e->w[i] = e->wg[i] = nullptr; e->b[i] = nullptr; }
    for (int i = 0; i < n_conv; ++i) { e->w[i] = e->wg[i] = weights[2 * i]; e->b[i] = weights[2 * i + 1]; }
    e->arith = MFTX_ARITH_F32;
    e->use_graph = 1;
    e->graphs = new (std::nothrow) GraphCache;
    *out = e;
    return 0;
}

extern "C" int mftx_encoder_set_split_weights(mftx_encoder *e, const void *const *split, int n) {
    if (e && e->magic == ENC_MAGIC && e->graphs) e->graphs->clear();
    if (!e || e->magic != ENC_MAGIC) return fail(MFTX_E_STATE, "encoder_set_split_weights: bad handle");
    const int n_conv = e->instance_norm ? EC_COUNT - 1 : EC_COUNT;
    if (!split) {                                    // back to fp32 MFMA
        for (int i = 0; i < n_conv; ++i) e->wg[i] = e->w[i];
        e->arith = MFTX_ARITH_F32;
        return 0;
    }
    if (n != n_conv) return fail(MFTX_E_ARG, "encoder_set_split_weights: expected %d weights, got %d", n_conv, n);
    for (int i = 0; i < n_conv; ++i)
        if (!split[i] || !aligned16(split[i])) return fail(MFTX_E_ALIGN, "encoder_set_split_weights: weight %d null or unaligned", i);
    for (int i = 0; i < n_conv; ++i) e->wg[i] = static_cast<const float *>(split[i]);
    e->arith = MFTX_ARITH_SPLIT;
    return 0;
}

extern "C" void mftx_encoder_destroy(mftx_encoder *e) {
    if (e && e->magic == ENC_MAGIC) { e->magic = 0; delete e->graphs; delete e; }
}

extern "C" int mftx_encoder_set_graph(mftx_encoder *e, int on) {
    if (!e || e->magic != ENC_MAGIC) return fail(MFTX_E_STATE, "encoder_set_graph: bad handle");
    e->use_graph = on ? 1 : 0;
    return 0;
}

extern "C" size_t mftx_encoder_workspace_bytes(int H0, int W0) {
    if (H0 <= 0 || W0 <= 0) return 0;
    const int Hp = (H0 + 7) / 8 * 8, Wp = (W0 + 7) / 8 * 8;
    return enc_carve(nullptr, Hp, Wp).bytes;
}

#define TRY(expr) do { int _e = (expr); if (_e) return _e; } while (0)

namespace {
struct Enc {
    const mftx_encoder *e;
    EncWs ws;
    hipStream_t s;

    // a_split / out_split (split arithmetic only): the input is / the output is written in split form
    int conv(int slot, const float *in, int cin, int lda, int hin, int win, float *out, int cout, int ldo, int h,
             int w, int k, int stride, int act, const float *residual = nullptr, bool a_split = false, bool out_split = false) {
        mftx_conv_desc d{};
        d.a0 = in; d.lda0 = lda; d.c0 = cin;
        d.wpk = e->wg[slot]; d.bias = e->b[slot]; d.arith = e->arith;
        d.a_split = SP() && a_split; d.out_split = SP() && out_split;
        d.out = out; d.ldo = ldo; d.P = 1; d.h = h; d.w = w; d.N = cout; d.kh = k; d.kw = k;
        d.act = act; d.out_scale = 1.f;
        d.stride = stride; d.hin = hin; d.win = win;
        if (residual) { d.addend = residual; d.ld_addend = cout; d.residual_mode = 1; }
        ProfConvCat cat(PC_ENC_GEMM);
        return launch_conv(d, s);
    }
    bool SP() const { return e->arith == MFTX_ARITH_SPLIT; }
    int norm(float *x, int rows, int C, int mode, const float *res = nullptr) {
        {
            ProfScope prof(PC_ENC_NORM, s, 4.0 * rows * C);
            hipLaunchKernelGGL(instnorm_partial_kernel, dim3(IN_SLABS), dim3(256), 0, s, x, rows, C, ws.part);
        }
        TRY(check_launch("instnorm_partial"));
        {
            ProfScope prof(PC_ENC_NORM, s, 0);
            hipLaunchKernelGGL(instnorm_finalize_kernel, dim3(cdiv(C, 16)), dim3(256), 0, s, ws.part, rows, C, ws.stat);
        }
        TRY(check_launch("instnorm_finalize"));
        const long long n8 = (long long)rows * C / 8;
        const int blocks = (int)std::min<long long>((n8 + 255) / 256, 4096);
        {
            ProfScope prof(PC_ENC_NORM, s, (mode == 1 ? 12.0 : 8.0) * rows * C);
            hipLaunchKernelGGL(instnorm_apply_kernel, dim3(blocks), dim3(256), 0, s, x, rows, C, ws.stat, res, mode, SP() ? 1 : 0);
        }
        return check_launch("instnorm_apply");
    }
    // one residual block (core/extractor.py:6-62); x: [hin*win][cin] -> out: [h*w][planes]; tmp, sc scratch
    int block(int c1, int c2, int ds, const float *x, int cin, int hin, int win, float *tmp, float *sc, float *out,
              int planes, int h, int w, int stride) {
        const int rows = h * w;
        if (e->instance_norm) {
            // (split arithmetic: every normalised map is in split form -- written so by instnorm_apply -- and every convolution here
            // reads such a map; the convolutions' own outputs stay fp32 for the statistics)
            TRY(conv(c1, x, cin, cin, hin, win, tmp, planes, planes, h, w, 3, stride, 0, nullptr, true));
            TRY(norm(tmp, rows, planes, 0));
            TRY(conv(c2, tmp, planes, planes, h, w, out, planes, planes, h, w, 3, 1, 0, nullptr, true));
            const float *shortcut = x;
            if (stride != 1) {
                TRY(conv(ds, x, cin, cin, hin, win, sc, planes, planes, h, w, 1, stride, 0, nullptr, true));
                TRY(norm(sc, rows, planes, 2));
                shortcut = sc;
            }
            return norm(out, rows, planes, 1, shortcut);
        }
        // batch norm folded: conv+bias+relu, then relu(x + relu(conv+bias)); the block's inner map goes to its second convolution in
        // split form (its outputs stay fp32: the next block adds them as they are)
        TRY(conv(c1, x, cin, cin, hin, win, tmp, planes, planes, h, w, 3, stride, 1, nullptr, false, true));
        const float *shortcut = x;
        if (stride != 1) {
            TRY(conv(ds, x, cin, cin, hin, win, sc, planes, planes, h, w, 1, stride, 0));
            shortcut = sc;
        }
        return conv(c2, tmp, planes, planes, h, w, out, planes, planes, h, w, 3, 1, 1, shortcut, true);
    }
};
}  // namespace

// img: uint8 [H0][W0][3] BGR on the device.  fnet: out0 = fmap [h*w][256].  cnet: out0 = net
// [h*w][128] (tanh), out1 = inp [h*w][128] (relu); h = Hp/8, w = Wp/8.
extern "C" int mftx_encoder_forward(mftx_encoder *e, const uint8_t *img, int H0, int W0, float *out0, float *out1,
                                    void *workspace, size_t workspace_bytes, void *stream) {
    if (!e || e->magic != ENC_MAGIC) return fail(MFTX_E_STATE, "encoder_forward: bad handle");
    if (!img || !out0 || !workspace || (!e->instance_norm && !out1)) return fail(MFTX_E_ARG, "encoder_forward: null pointer");
    if (H0 < 16 || W0 < 16) return fail(MFTX_E_ARG, "encoder_forward: image too small");
    if (reinterpret_cast<uintptr_t>(workspace) & 255) return fail(MFTX_E_ALIGN, "encoder_forward: workspace must be 256-byte aligned");
    const int ph = (((H0 / 8) + 1) * 8 - H0) % 8, pw = (((W0 / 8) + 1) * 8 - W0) % 8;
    const int Hp = H0 + ph, Wp = W0 + pw, pl = pw / 2, pt = ph / 2;
    Enc E{e, enc_carve(workspace, Hp, Wp), (hipStream_t)stream};
    if (E.ws.bytes > workspace_bytes) return fail(MFTX_E_WORKSPACE, "encoder_forward: workspace %zu < %zu", workspace_bytes, E.ws.bytes);
    const bool graphs_on = e->graphs && e->use_graph && !prof_enabled();
    bool proxied = false;
    if (graphs_on && E.s == nullptr) {               // the legacy stream (PyTorch's default) cannot be captured: graph_cache.h
        hipStream_t own = e->graphs->proxy.enter(nullptr);
        if (own) { E.s = own; proxied = true; }
    }
    struct Leave { GraphCache *g; bool on; ~Leave() { if (on) g->proxy.leave(nullptr); } } leave{e->graphs, proxied};
    hipStream_t s = E.s;
    {
        ProfScope prof(PC_GLUE, s, 0);
        hipLaunchKernelGGL(enc_prep_kernel, dim3(cdiv(Wp + 6, 256), Hp), dim3(256), 0, s, img, H0, W0, pl, pt, Hp, Wp, E.ws.img);
    }
    TRY(check_launch("enc_prep"));
    // From the stem to the last residual block the layers touch the workspace only: one hipGraph per (size, workspace,
    // stream), see graph_cache.h; the pre-processing kernel (reads the caller's image) and the head (writes the caller's
    // maps) are launched plainly around it.
    const int h3 = Hp / 8, w3 = Wp / 8;
    auto body = [&]() -> int {
    // stem: 7 taps (rows) x 28-float windows starting at padded column 2x: kh = 7, kw = 1, no x padding
    const int h1 = Hp / 2, w1 = Wp / 2;
    {
        mftx_conv_desc d{};
        d.a0 = E.ws.img; d.lda0 = 4; d.c0 = 28;
        d.wpk = e->wg[EC_STEM]; d.bias = e->b[EC_STEM]; d.arith = e->arith;
        d.out = E.ws.a; d.ldo = 64; d.P = 1; d.h = h1; d.w = w1; d.N = 64; d.kh = 7; d.kw = 1;
        d.act = e->instance_norm ? 0 : 1; d.out_scale = 1.f;
        d.stride = 2; d.hin = Hp; d.win = Wp + 6; d.pad_y = 3; d.pad_x = -1;
        ProfConvCat cat(PC_ENC_GEMM);
        TRY(launch_conv(d, s));
    }
    if (e->instance_norm) TRY(E.norm(E.ws.a, h1 * w1, 64, 0));
    // stage 1 (64, stride 1): a -> c -> a
    TRY(E.block(EC_L1B0C1, EC_L1B0C2, -1, E.ws.a, 64, h1, w1, E.ws.b, nullptr, E.ws.c, 64, h1, w1, 1));
    TRY(E.block(EC_L1B1C1, EC_L1B1C2, -1, E.ws.c, 64, h1, w1, E.ws.b, nullptr, E.ws.a, 64, h1, w1, 1));
    // stage 2 (96, stride 2)
    const int h2 = h1 / 2, w2 = w1 / 2;
    float *sc2 = E.ws.b + (size_t)h2 * w2 * 96;      // tmp and shortcut share buffer b (both quarter-size maps)
    TRY(E.block(EC_L2B0C1, EC_L2B0C2, EC_L2B0DS, E.ws.a, 64, h1, w1, E.ws.b, sc2, E.ws.c, 96, h2, w2, 2));
    TRY(E.block(EC_L2B1C1, EC_L2B1C2, -1, E.ws.c, 96, h2, w2, E.ws.b, nullptr, E.ws.a, 96, h2, w2, 1));
    // stage 3 (128, stride 2)
    float *sc3 = E.ws.b + (size_t)h3 * w3 * 128;
    TRY(E.block(EC_L3B0C1, EC_L3B0C2, EC_L3B0DS, E.ws.a, 96, h2, w2, E.ws.b, sc3, E.ws.c, 128, h3, w3, 2));
    return E.block(EC_L3B1C1, EC_L3B1C2, -1, E.ws.c, 128, h3, w3, E.ws.b, nullptr, E.ws.a, 128, h3, w3, 1);
    };   // body
    if (graphs_on) {
        GraphKey key{};
        key.v[0] = (uintptr_t)H0; key.v[1] = (uintptr_t)W0; key.v[2] = reinterpret_cast<uintptr_t>(workspace);
        key.v[3] = reinterpret_cast<uintptr_t>(s); key.v[4] = (uintptr_t)e->arith;
        TRY(e->graphs->run(key, s, body));
    } else {
        TRY(body());
    }
    // head
    if (e->instance_norm) return E.conv(EC_HEAD, E.ws.a, 128, 128, h3, w3, out0, 256, 256, h3, w3, 1, 1, 0, nullptr, true);
    TRY(E.conv(EC_HEAD, E.ws.a, 128, 128, h3, w3, out0, 128, 128, h3, w3, 1, 1, 3));   // net = tanh(first 128)
    return E.conv(EC_HEAD2, E.ws.a, 128, 128, h3, w3, out1, 128, 128, h3, w3, 1, 1, 1);  // inp = relu(last 128)
}
