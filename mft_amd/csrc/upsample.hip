// Convex 8x upsampling of flow / occlusion logits / log-variance with the
// post-processing of the flow wrapper fused in (HBM-bound).
//
// core/raft.py:83-94:  m = softmax_k(mask[k*64 + sy*8 + sx]),  k = ky*3 + kx
//   out[c, 8y+sy, 8x+sx] = sum_k m_k * mult * x[c, y+ky-1, x+kx-1]   (zeros outside)
// with mult = 8 for flow, 1 for the OU heads (core/raft.py:190,211,218); then
// MFT/raft.py:57-62: unpad, occl = softmax(logits)[1], sigma = sqrt(exp(u)).
//
// One thread per full-resolution pixel; a wave covers 64 consecutive x so the
// planar stores are fully coalesced and each coarse cell's mask block is read
// as 8 consecutive floats per k.  The softmax over the 9 mask logits is
// computed once and shared by the 5 upsampled channels.
#include "common.h"
#include "profile.h"

namespace mftx {

struct UpArgs {
    const float *flow_lr;  // [M][2]
    const float *ou;       // [M][ld_ou]: logit0, logit1, log-variance
    int ld_ou;
    const float *mask;     // [M][576]
    int P, h, w;
    int pl, pt;            // left / top crop
    int H0, W0;            // unpadded output size
    float *flow, *occl, *sigma;
};

__global__ __launch_bounds__(256) void convex_upsample_kernel(UpArgs p) {
    const int X0 = blockIdx.x * blockDim.x + threadIdx.x;   // output (unpadded) x
    const int Y0 = blockIdx.y;
    const int img = blockIdx.z;
    if (X0 >= p.W0) return;
    const int X = X0 + p.pl, Y = Y0 + p.pt;                 // padded coordinates
    const int y = Y >> 3, sy = Y & 7, x = X >> 3, sx = X & 7;
    const long long cell = ((long long)img * p.h + y) * p.w + x;
    const float *mk = p.mask + cell * 576 + sy * 8 + sx;
    float m[9];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 9; ++k) { m[k] = mk[k * 64]; mx = fmaxf(mx, m[k]); }
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { m[k] = expf(m[k] - mx); den += m[k]; }
    float fxv = 0.f, fyv = 0.f, l0 = 0.f, l1 = 0.f, u = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int ny = y + k / 3 - 1, nx = x + k % 3 - 1;
        if (ny < 0 || ny >= p.h || nx < 0 || nx >= p.w) continue;
        const long long nc = ((long long)img * p.h + ny) * p.w + nx;
        const float wk = m[k] / den;
        const float2 f = reinterpret_cast<const float2 *>(p.flow_lr)[nc];
        const float *o = p.ou + nc * p.ld_ou;
        fxv += wk * (8.f * f.x);
        fyv += wk * (8.f * f.y);
        l0 += wk * o[0];
        l1 += wk * o[1];
        u += wk * o[2];
    }
    const long long plane = (long long)p.H0 * p.W0;
    const long long pix = (long long)Y0 * p.W0 + X0;
    p.flow[(img * 2 + 0) * plane + pix] = fxv;
    p.flow[(img * 2 + 1) * plane + pix] = fyv;
    // softmax over the two logits, channel 1
    const float lm = fmaxf(l0, l1);
    const float e0 = expf(l0 - lm), e1 = expf(l1 - lm);
    p.occl[img * plane + pix] = e1 / (e0 + e1);
    p.sigma[img * plane + pix] = sqrtf(expf(u));
}

int launch_convex_upsample(const float *flow_lr, const float *ou, int ld_ou, const float *mask, int P, int h,
                           int w, int pl, int pr, int pt, int pb, float *flow, float *occl, float *sigma,
                           hipStream_t s) {
    UpArgs a;
    a.flow_lr = flow_lr; a.ou = ou; a.ld_ou = ld_ou; a.mask = mask;
    a.P = P; a.h = h; a.w = w; a.pl = pl; a.pt = pt;
    a.H0 = 8 * h - pt - pb; a.W0 = 8 * w - pl - pr;
    a.flow = flow; a.occl = occl; a.sigma = sigma;
    if (a.H0 <= 0 || a.W0 <= 0) return fail(MFTX_E_ARG, "convex_upsample: bad padding");
    dim3 grid(cdiv(a.W0, 256), a.H0, P);
    ProfScope prof(PC_UPSAMPLE, s, (double)P * h * w * (576 + 5) * 4 + (double)P * 4 * a.H0 * a.W0 * 4);
    hipLaunchKernelGGL(convex_upsample_kernel, grid, dim3(256), 0, s, a);
    return check_launch("convex_upsample");
}

}  // namespace mftx
