// MFT flow chaining and per-pixel best-chain selection (HBM-bound).
//
// chain (MFT/MFT.py:233-239, MFT/results.py:87-136), per pixel g = (x, y):
//   p     = g + flowL
//   B(X)  = bilinear sample of X at p, zeros outside, align_corners=True, after
//           the reference's normalise (MFT/utils/interpolation.py:69-72) /
//           un-normalise round trip, reproduced here in fp32
//   flow  = (p + B(flowR)) - g ;  occl = max(occlL, B(occlR)) ;
//   sigma = sqrt(sigmaL^2 + B(sigmaR)^2)
// select (MFT/MFT.py:112-143, MFT/results.py:250-265): first arg-max over the
// candidates of -sigma, with -inf where occl > thr; then occl = 1 where the
// selected flow leaves the image.
//
// The reference does 3 grid_samples + ~25 launches per candidate and a
// stack/max/gather pass; here one thread owns a pixel, walks the K candidates,
// and keeps the running best in registers: (32 K + 16) B/pixel of traffic.
// `chain_px` is shared by all three kernels so that chain -> all-gather ->
// select (multi-GPU) is bitwise identical to the fused single-GPU kernel.
// Compiled with -ffp-contract=off: no FMA contraction, products and sums round
// exactly as written.
#include "common.h"
#include "profile.h"
#include <initializer_list>

namespace mftx {

struct Planes { const float *flow, *occl, *sigma; };
struct Chained { float fx, fy, occ, sig; };

__device__ __forceinline__ float tap(const float *pl, int H, int W, int yy, int xx) {
    return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? pl[(long long)yy * W + xx] : 0.f;
}

// NaN handling follows torch: a NaN sampled occlusion propagates through
// maximum(); fmaxf would drop it, so patch it up explicitly.
__device__ __forceinline__ float max_nanprop(float a, float b) {
    return (a != a || b != b) ? NAN : fmaxf(a, b);
}

__device__ __forceinline__ Chained chain_px(const Planes &L, const Planes &R, int H, int W, int x, int y,
                                           float sx, float sy) {
    const long long pix = (long long)y * W + x;
    const long long plane = (long long)H * W;
    const float gx = (float)x, gy = (float)y;
    const float px = gx + L.flow[pix];
    const float py = gy + L.flow[plane + pix];
    // normalise: p * float32(2/(W-1)) - 1 ; un-normalise: ((g + 1) / 2) * (W - 1)
    const float ix = ((px * sx - 1.f) + 1.f) / 2.f * (float)(W - 1);
    const float iy = ((py * sy - 1.f) + 1.f) / 2.f * (float)(H - 1);
    const float flx = floorf(ix), fly = floorf(iy);
    const float wx = ix - flx, wy = iy - fly;
    const int x0 = (int)fminf(fmaxf(flx, -1.0e6f), 1.0e6f);
    const int y0 = (int)fminf(fmaxf(fly, -1.0e6f), 1.0e6f);
    const float w00 = (1.f - wx) * (1.f - wy), w01 = wx * (1.f - wy), w10 = (1.f - wx) * wy, w11 = wx * wy;
    auto samp = [&](const float *pl) {
        return tap(pl, H, W, y0, x0) * w00 + tap(pl, H, W, y0, x0 + 1) * w01 +
               tap(pl, H, W, y0 + 1, x0) * w10 + tap(pl, H, W, y0 + 1, x0 + 1) * w11;
    };
    Chained c;
    c.fx = (px + samp(R.flow)) - gx;
    c.fy = (py + samp(R.flow + plane)) - gy;
    c.occ = max_nanprop(L.occl[pix], samp(R.occl));
    const float sl = L.sigma[pix], sr = samp(R.sigma);
    c.sig = sqrtf(sl * sl + sr * sr);
    return c;
}

__global__ __launch_bounds__(256) void chain_kernel(Planes L, Planes R, int H, int W, float sx, float sy,
                                                    float *flowO, float *occlO, float *sigmaO) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W) return;
    const Chained c = chain_px(L, R, H, W, x, y, sx, sy);
    const long long pix = (long long)y * W + x, plane = (long long)H * W;
    flowO[pix] = c.fx; flowO[plane + pix] = c.fy; occlO[pix] = c.occ; sigmaO[pix] = c.sig;
}

// warp_backward (MFT/results.py:116-136): out[c] = B(img[c]) at p = g + flow
__global__ __launch_bounds__(256) void warp_backward_kernel(const float *__restrict__ flow,
                                                            const float *__restrict__ img, int C, int H, int W,
                                                            float sx, float sy, float *__restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W) return;
    const long long pix = (long long)y * W + x, plane = (long long)H * W;
    const float px = (float)x + flow[pix];
    const float py = (float)y + flow[plane + pix];
    const float ix = ((px * sx - 1.f) + 1.f) / 2.f * (float)(W - 1);
    const float iy = ((py * sy - 1.f) + 1.f) / 2.f * (float)(H - 1);
    const float flx = floorf(ix), fly = floorf(iy);
    const float wx = ix - flx, wy = iy - fly;
    const int x0 = (int)fminf(fmaxf(flx, -1.0e6f), 1.0e6f);
    const int y0 = (int)fminf(fmaxf(fly, -1.0e6f), 1.0e6f);
    const float w00 = (1.f - wx) * (1.f - wy), w01 = wx * (1.f - wy), w10 = (1.f - wx) * wy, w11 = wx * wy;
    for (int c = 0; c < C; ++c) {
        const float *pl = img + c * plane;
        out[c * plane + pix] = tap(pl, H, W, y0, x0) * w00 + tap(pl, H, W, y0, x0 + 1) * w01 +
                               tap(pl, H, W, y0 + 1, x0) * w10 + tap(pl, H, W, y0 + 1, x0 + 1) * w11;
    }
}

struct CandSet {
    int K;
    Planes L[MFTX_MAX_CANDIDATES];
    Planes R[MFTX_MAX_CANDIDATES];
};

struct Best { float score; Chained c; int k; };

__device__ __forceinline__ void consider(Best &b, const Chained &c, int k, float thr) {
    // torch: scores = -sigma; scores[occl > thr] = -inf; max(dim=0) keeps the
    // first maximal index (a NaN score wins over everything, first NaN kept).
    const float score = (c.occ > thr) ? -INFINITY : -c.sig;
    bool take;
    if (k == 0) take = true;
    else if (b.score != b.score) take = false;          // current best is NaN
    else if (score != score) take = true;               // NaN beats numbers
    else take = score > b.score;                        // strict: first max wins
    if (take) { b.score = score; b.c = c; b.k = k; }
}

__device__ __forceinline__ void write_selected(const Best &b, int H, int W, int x, int y, float *flowO,
                                               float *occlO, float *sigmaO, int8_t *chosen) {
    const long long pix = (long long)y * W + x, plane = (long long)H * W;
    const float qx = (float)x + b.c.fx, qy = (float)y + b.c.fy;
    const bool invalid = (qx < 0.f) | (qy < 0.f) | (qx >= (float)W) | (qy >= (float)H);
    flowO[pix] = b.c.fx;
    flowO[plane + pix] = b.c.fy;
    occlO[pix] = invalid ? 1.f : b.c.occ;
    sigmaO[pix] = b.c.sig;
    if (chosen) chosen[pix] = (int8_t)b.k;
}

__global__ __launch_bounds__(256) void chain_select_kernel(CandSet cs, float thr, int H, int W, float sx,
                                                           float sy, float *flowO, float *occlO, float *sigmaO,
                                                           int8_t *chosen) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W) return;
    Best b;
    b.score = 0.f; b.k = 0; b.c = Chained{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < cs.K; ++k) consider(b, chain_px(cs.L[k], cs.R[k], H, W, x, y, sx, sy), k, thr);
    write_selected(b, H, W, x, y, flowO, occlO, sigmaO, chosen);
}

struct SelSet {
    int K;
    Planes C[MFTX_MAX_CANDIDATES];
};

__global__ __launch_bounds__(256) void select_kernel(SelSet ss, float thr, int H, int W, float *flowO,
                                                     float *occlO, float *sigmaO, int8_t *chosen) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W) return;
    const long long pix = (long long)y * W + x, plane = (long long)H * W;
    Best b;
    b.score = 0.f; b.k = 0; b.c = Chained{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < ss.K; ++k) {
        Chained c;
        c.fx = ss.C[k].flow[pix]; c.fy = ss.C[k].flow[plane + pix];
        c.occ = ss.C[k].occl[pix]; c.sig = ss.C[k].sigma[pix];
        consider(b, c, k, thr);
    }
    write_selected(b, H, W, x, y, flowO, occlO, sigmaO, chosen);
}

// ---- PACKED right operands ([H][W][4] = fx, fy, occl, sigma per pixel, written by the upsampler): each bilinear
// tap is ONE 16-byte gather instead of four 4-byte ones.  Same arithmetic, operation by operation, as chain_px
// (bitwise equal results).
struct PackedSet {
    int K;
    Planes L[MFTX_MAX_CANDIDATES];
    const float4 *R[MFTX_MAX_CANDIDATES];
};

__device__ __forceinline__ void write_selected4(const Best (&b)[4], int H, int W, int x, int y, float *flowO,
                                                float *occlO, float *sigmaO, int8_t *chosen) {
    const long long pix = (long long)y * W + x, plane = (long long)H * W;
    float ox[4], oy[4], oo[4], os[4];
    char4 ck;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float qx = (float)(x + i) + b[i].c.fx, qy = (float)y + b[i].c.fy;
        const bool invalid = (qx < 0.f) | (qy < 0.f) | (qx >= (float)W) | (qy >= (float)H);
        ox[i] = b[i].c.fx; oy[i] = b[i].c.fy; oo[i] = invalid ? 1.f : b[i].c.occ; os[i] = b[i].c.sig;
    }
    ck.x = (signed char)b[0].k; ck.y = (signed char)b[1].k; ck.z = (signed char)b[2].k; ck.w = (signed char)b[3].k;
    *reinterpret_cast<float4 *>(flowO + pix) = make_float4(ox[0], ox[1], ox[2], ox[3]);
    *reinterpret_cast<float4 *>(flowO + plane + pix) = make_float4(oy[0], oy[1], oy[2], oy[3]);
    *reinterpret_cast<float4 *>(occlO + pix) = make_float4(oo[0], oo[1], oo[2], oo[3]);
    *reinterpret_cast<float4 *>(sigmaO + pix) = make_float4(os[0], os[1], os[2], os[3]);
    if (chosen) *reinterpret_cast<char4 *>(chosen + pix) = ck;
}

// One pixel per thread, right operands PACKED: 4 coalesced 4-byte loads of the left planes + 4 sixteen-byte gathers
// per candidate instead of 4 + 16 four-byte gathers (the planar kernel is bound by the texture-address rate of
// its gathers).  The chain of a candidate is two dependent memory round trips (left planes -> tap addresses -> taps);
// walked candidate by candidate that is 2 K round trips per thread and the kernel is latency-bound (24 us for 63 MB at
// K = 7).  With K known at compile time (KT = 1..8) ALL left planes are loaded first, then the taps in batches of
// four candidates: 1 + ceil(K / 4) round trips.  Taps outside the image are read at a clamped address and replaced by
// zeros afterwards (no branches).  Same arithmetic, operation by operation, as chain_px: bitwise equal results.
struct ChainPrep { float px, py, gx, gy, w00, w01, w10, w11; int x0, y0; };

__device__ __forceinline__ ChainPrep chain_prep(float lfx, float lfy, int H, int W, int x, int y, float sx, float sy) {
    ChainPrep c;
    c.gx = (float)x; c.gy = (float)y;
    c.px = c.gx + lfx;
    c.py = c.gy + lfy;
    const float ix = ((c.px * sx - 1.f) + 1.f) / 2.f * (float)(W - 1);
    const float iy = ((c.py * sy - 1.f) + 1.f) / 2.f * (float)(H - 1);
    const float flx = floorf(ix), fly = floorf(iy);
    const float wx = ix - flx, wy = iy - fly;
    c.x0 = (int)fminf(fmaxf(flx, -1.0e6f), 1.0e6f);
    c.y0 = (int)fminf(fmaxf(fly, -1.0e6f), 1.0e6f);
    c.w00 = (1.f - wx) * (1.f - wy); c.w01 = wx * (1.f - wy); c.w10 = (1.f - wx) * wy; c.w11 = wx * wy;
    return c;
}

__device__ __forceinline__ float4 packed_tap(const float4 *R, int H, int W, int yy, int xx) {
    const bool ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W);
    const int cy = min(max(yy, 0), H - 1), cx = min(max(xx, 0), W - 1);
    const float4 v = R[(long long)cy * W + cx];
    return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
}

__device__ __forceinline__ Chained chain_finish(const ChainPrep &c, float locc, float lsig, const float4 &a, const float4 &b,
                                                const float4 &cc, const float4 &d) {
    Chained o;
    o.fx = (c.px + (a.x * c.w00 + b.x * c.w01 + cc.x * c.w10 + d.x * c.w11)) - c.gx;
    o.fy = (c.py + (a.y * c.w00 + b.y * c.w01 + cc.y * c.w10 + d.y * c.w11)) - c.gy;
    o.occ = max_nanprop(locc, a.z * c.w00 + b.z * c.w01 + cc.z * c.w10 + d.z * c.w11);
    const float sr = a.w * c.w00 + b.w * c.w01 + cc.w * c.w10 + d.w * c.w11;
    o.sig = sqrtf(lsig * lsig + sr * sr);
    return o;
}

template <int KT>    // KT = number of candidates (1..8), or 0: any K, candidate by candidate
__global__ __launch_bounds__(256) void chain_select_packed_kernel(PackedSet ps, float thr, int H, int W, float sx,
                                                                  float sy, float *flowO, float *occlO, float *sigmaO,
                                                                  int8_t *chosen) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= W) return;
    const long long pix = (long long)y * W + x, plane = (long long)H * W;
    Best b;
    b.score = 0.f; b.k = 0; b.c = Chained{0.f, 0.f, 0.f, 0.f};
    if constexpr (KT == 0) {
        for (int k = 0; k < ps.K; ++k) {
            const Planes L = ps.L[k];
            const ChainPrep c = chain_prep(L.flow[pix], L.flow[plane + pix], H, W, x, y, sx, sy);
            const float4 *R = ps.R[k];
            consider(b, chain_finish(c, L.occl[pix], L.sigma[pix], packed_tap(R, H, W, c.y0, c.x0), packed_tap(R, H, W, c.y0, c.x0 + 1),
                                     packed_tap(R, H, W, c.y0 + 1, c.x0), packed_tap(R, H, W, c.y0 + 1, c.x0 + 1)), k, thr);
        }
    } else {
        float lfx[KT], lfy[KT], loc[KT], lsg[KT];
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            lfx[k] = ps.L[k].flow[pix]; lfy[k] = ps.L[k].flow[plane + pix];
            loc[k] = ps.L[k].occl[pix]; lsg[k] = ps.L[k].sigma[pix];
        }
#pragma unroll
        for (int k0 = 0; k0 < KT; k0 += 4) {
            constexpr int B = 4;
            ChainPrep c[B];
            float4 t[B][4];
#pragma unroll
            for (int j = 0; j < B; ++j) {
                if (k0 + j < KT) {
                    c[j] = chain_prep(lfx[k0 + j], lfy[k0 + j], H, W, x, y, sx, sy);
                    const float4 *R = ps.R[k0 + j];
                    t[j][0] = packed_tap(R, H, W, c[j].y0, c[j].x0);
                    t[j][1] = packed_tap(R, H, W, c[j].y0, c[j].x0 + 1);
                    t[j][2] = packed_tap(R, H, W, c[j].y0 + 1, c[j].x0);
                    t[j][3] = packed_tap(R, H, W, c[j].y0 + 1, c[j].x0 + 1);
                }
            }
#pragma unroll
            for (int j = 0; j < B; ++j)
                if (k0 + j < KT)
                    consider(b, chain_finish(c[j], loc[k0 + j], lsg[k0 + j], t[j][0], t[j][1], t[j][2], t[j][3]), k0 + j, thr);
        }
    }
    write_selected(b, H, W, x, y, flowO, occlO, sigmaO, chosen);
}

__global__ __launch_bounds__(256) void select4_kernel(SelSet ss, float thr, int H, int W, float *flowO, float *occlO,
                                                      float *sigmaO, int8_t *chosen) {
    const int x = 4 * (blockIdx.x * blockDim.x + threadIdx.x);
    const int y = blockIdx.y;
    if (x >= W) return;
    const long long pix = (long long)y * W + x, plane = (long long)H * W;
    Best b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { b[i].score = 0.f; b[i].k = 0; b[i].c = Chained{0.f, 0.f, 0.f, 0.f}; }
    for (int k = 0; k < ss.K; ++k) {
        const float4 fx = *reinterpret_cast<const float4 *>(ss.C[k].flow + pix);
        const float4 fy = *reinterpret_cast<const float4 *>(ss.C[k].flow + plane + pix);
        const float4 oc = *reinterpret_cast<const float4 *>(ss.C[k].occl + pix);
        const float4 sg = *reinterpret_cast<const float4 *>(ss.C[k].sigma + pix);
        consider(b[0], Chained{fx.x, fy.x, oc.x, sg.x}, k, thr);
        consider(b[1], Chained{fx.y, fy.y, oc.y, sg.y}, k, thr);
        consider(b[2], Chained{fx.z, fy.z, oc.z, sg.z}, k, thr);
        consider(b[3], Chained{fx.w, fy.w, oc.w, sg.w}, k, thr);
    }
    write_selected4(b, H, W, x, y, flowO, occlO, sigmaO, chosen);
}

}  // namespace mftx

using namespace mftx;

// float4 access to every plane: W % 4 == 0 and all bases 16-byte aligned
static bool vec4_ok(int W, std::initializer_list<const void *> ptrs) {
    if (W % 4) return false;
    for (const void *p : ptrs) if (p != nullptr && !aligned16(p)) return false;
    return true;
}

static void scales(int H, int W, float &sx, float &sy) {
    // np.array([2/(W-1), 2/(H-1)]).astype(np.float32)  (double division, then rounded)
    sx = (float)(2.0 / (double)(W - 1));
    sy = (float)(2.0 / (double)(H - 1));
}

extern "C" int mftx_chain(const float *flowL, const float *occlL, const float *sigmaL, const float *flowR,
                          const float *occlR, const float *sigmaR, int H, int W, float *flowO, float *occlO,
                          float *sigmaO, void *stream) {
    if (!flowL || !occlL || !sigmaL || !flowR || !occlR || !sigmaR || !flowO || !occlO || !sigmaO)
        return fail(MFTX_E_ARG, "chain: null pointer");
    if (H < 2 || W < 2) return fail(MFTX_E_ARG, "chain: H and W must be >= 2");
    float sx, sy;
    scales(H, W, sx, sy);
    Planes L{flowL, occlL, sigmaL}, R{flowR, occlR, sigmaR};
    ProfScope prof(PC_CHAIN, (hipStream_t)stream, 48.0 * H * W);
    hipLaunchKernelGGL(chain_kernel, dim3(cdiv(W, 256), H), dim3(256), 0, (hipStream_t)stream, L, R, H, W, sx, sy,
                       flowO, occlO, sigmaO);
    return check_launch("chain");
}

extern "C" int mftx_warp_backward(const float *flow, const float *img, int C, int H, int W, float *out,
                                  void *stream) {
    if (!flow || !img || !out) return fail(MFTX_E_ARG, "warp_backward: null pointer");
    if (C < 1 || H < 2 || W < 2) return fail(MFTX_E_ARG, "warp_backward: need C >= 1, H, W >= 2");
    float sx, sy;
    scales(H, W, sx, sy);
    hipLaunchKernelGGL(warp_backward_kernel, dim3(cdiv(W, 256), H), dim3(256), 0, (hipStream_t)stream, flow, img,
                       C, H, W, sx, sy, out);
    return check_launch("warp_backward");
}

extern "C" int mftx_select(int K, const float *const *flow, const float *const *occl, const float *const *sigma,
                           float thr, int H, int W, float *flowO, float *occlO, float *sigmaO, int8_t *chosen,
                           void *stream) {
    if (K < 1 || K > MFTX_MAX_CANDIDATES) return fail(MFTX_E_ARG, "select: K must be in 1..%d", MFTX_MAX_CANDIDATES);
    if (!flow || !occl || !sigma || !flowO || !occlO || !sigmaO) return fail(MFTX_E_ARG, "select: null pointer");
    if (H < 1 || W < 1) return fail(MFTX_E_ARG, "select: bad size");
    SelSet ss;
    ss.K = K;
    for (int k = 0; k < K; ++k) {
        if (!flow[k] || !occl[k] || !sigma[k]) return fail(MFTX_E_ARG, "select: null candidate %d", k);
        ss.C[k] = Planes{flow[k], occl[k], sigma[k]};
    }
    bool v4 = vec4_ok(W, {flowO, occlO, sigmaO}) && (chosen == nullptr || (reinterpret_cast<uintptr_t>(chosen) & 3) == 0);
    for (int k = 0; k < K && v4; ++k) v4 = vec4_ok(W, {flow[k], occl[k], sigma[k]});
    ProfScope prof(PC_CHAIN, (hipStream_t)stream, (16.0 * K + 16.0) * H * W);
    if (v4)
        hipLaunchKernelGGL(select4_kernel, dim3(cdiv(W, 256), H), dim3(64), 0, (hipStream_t)stream, ss, thr, H, W,
                           flowO, occlO, sigmaO, chosen);
    else
        hipLaunchKernelGGL(select_kernel, dim3(cdiv(W, 256), H), dim3(256), 0, (hipStream_t)stream, ss, thr, H, W,
                           flowO, occlO, sigmaO, chosen);
    return check_launch("select");
}

extern "C" int mftx_chain_select(int K, const float *const *flowL, const float *const *occlL,
                                 const float *const *sigmaL, const float *const *flowR,
                                 const float *const *occlR, const float *const *sigmaR, float thr, int H, int W,
                                 float *flowO, float *occlO, float *sigmaO, int8_t *chosen, void *stream) {
    if (K < 1 || K > MFTX_MAX_CANDIDATES)
        return fail(MFTX_E_ARG, "chain_select: K must be in 1..%d", MFTX_MAX_CANDIDATES);
    if (!flowL || !occlL || !sigmaL || !flowR || !occlR || !sigmaR || !flowO || !occlO || !sigmaO)
        return fail(MFTX_E_ARG, "chain_select: null pointer");
    if (H < 2 || W < 2) return fail(MFTX_E_ARG, "chain_select: H and W must be >= 2");
    CandSet cs;
    cs.K = K;
    for (int k = 0; k < K; ++k) {
        if (!flowL[k] || !occlL[k] || !sigmaL[k] || !flowR[k] || !occlR[k] || !sigmaR[k])
            return fail(MFTX_E_ARG, "chain_select: null candidate %d", k);
        cs.L[k] = Planes{flowL[k], occlL[k], sigmaL[k]};
        cs.R[k] = Planes{flowR[k], occlR[k], sigmaR[k]};
    }
    float sx, sy;
    scales(H, W, sx, sy);
    ProfScope prof(PC_CHAIN, (hipStream_t)stream, (32.0 * K + 16.0) * H * W);
    hipLaunchKernelGGL(chain_select_kernel, dim3(cdiv(W, 256), H), dim3(256), 0, (hipStream_t)stream, cs, thr, H,
                       W, sx, sy, flowO, occlO, sigmaO, chosen);
    return check_launch("chain_select");
}

extern "C" int mftx_chain_select_packed(int K, const float *const *flowL, const float *const *occlL,
                                        const float *const *sigmaL, const float *const *packedR, float thr, int H,
                                        int W, float *flowO, float *occlO, float *sigmaO, int8_t *chosen, void *stream) {
    if (K < 1 || K > MFTX_MAX_CANDIDATES)
        return fail(MFTX_E_ARG, "chain_select_packed: K must be in 1..%d", MFTX_MAX_CANDIDATES);
    if (!flowL || !occlL || !sigmaL || !packedR || !flowO || !occlO || !sigmaO)
        return fail(MFTX_E_ARG, "chain_select_packed: null pointer");
    if (H < 2 || W < 2) return fail(MFTX_E_ARG, "chain_select_packed: H and W must be >= 2");
    PackedSet ps;
    ps.K = K;
    for (int k = 0; k < K; ++k) {
        if (!flowL[k] || !occlL[k] || !sigmaL[k] || !packedR[k])
            return fail(MFTX_E_ARG, "chain_select_packed: null candidate %d", k);
        if (!aligned16(packedR[k])) return fail(MFTX_E_ALIGN, "chain_select_packed: packed operands must be 16-byte aligned");
        ps.L[k] = Planes{flowL[k], occlL[k], sigmaL[k]};
        ps.R[k] = reinterpret_cast<const float4 *>(packedR[k]);
    }
    float sx, sy;
    scales(H, W, sx, sy);
    ProfScope prof(PC_CHAIN, (hipStream_t)stream, (32.0 * K + 16.0) * H * W);
#define CSP_LAUNCH(KT)                                                                                               \
    hipLaunchKernelGGL(chain_select_packed_kernel<KT>, dim3(cdiv(W, 256), H), dim3(256), 0, (hipStream_t)stream, ps, thr, H, \
                       W, sx, sy, flowO, occlO, sigmaO, chosen)
    switch (K) {
        case 1: CSP_LAUNCH(1); break;
        case 2: CSP_LAUNCH(2); break;
        case 3: CSP_LAUNCH(3); break;
        case 4: CSP_LAUNCH(4); break;
        case 5: CSP_LAUNCH(5); break;
        case 6: CSP_LAUNCH(6); break;
        case 7: CSP_LAUNCH(7); break;
        case 8: CSP_LAUNCH(8); break;
        default: CSP_LAUNCH(0); break;
    }
#undef CSP_LAUNCH
    return check_launch("chain_select_packed");
}
