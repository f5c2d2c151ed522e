// Convex 8x upsampling of flow / occlusion logits / log-variance with the
// post-processing of the flow wrapper fused in (HBM-bound).
//
// core/raft.py:83-94:  m = softmax_k(mask[k*64 + sy*8 + sx]),  k = ky*3 + kx
//   out[c, 8y+sy, 8x+sx] = sum_k m_k * mult * x[c, y+ky-1, x+kx-1]   (zeros outside)
// with mult = 8 for flow, 1 for the OU heads (core/raft.py:190,211,218); then
// MFT/raft.py:57-62: unpad, occl = softmax(logits)[1], sigma = sqrt(exp(u)).
//
// One WAVE per coarse cell, lane = (sy, sx) of its 8 x 8 output pixels: the cell's 576 mask logits are
// channel k*64 + sy*8 + sx, so lane l reads mask[cell][k*64 + l] -- nine fully coalesced 256-byte loads, the
// 66 MB mask (7 pairs, 512 x 512) streams through once at line granularity.  The nine neighbours' coarse
// values (flow, two logits, log-variance) are the same for the whole wave: scalar loads.  The softmax over
// the 9 logits is computed once and shared by the 5 upsampled channels.  Outputs: planar flow / occlusion /
// sigma (the API's layout; 32-byte runs per plane and output row, merged into full lines in L2 with the
// neighbouring cells' runs) and, optionally, the same four values interleaved per pixel ("packed" FlowOU:
// fx, fy, occl, sigma) -- 128-byte runs, one 16-byte gather per bilinear tap for the chaining kernel.
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
    float *packed;         // optional [P][H0][W0][4]
    unsigned *nonfinite;   // optional device counter: += the number of output pixels with a non-finite value (see raft_engine.hip)
    int cells;
};

constexpr int UP_WAVES = 4;

__global__ __launch_bounds__(64 * UP_WAVES) void convex_upsample_kernel(UpArgs p) {
    const int lane = threadIdx.x & 63;
    const int cell = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * UP_WAVES + (threadIdx.x >> 6)));
    if (cell >= p.cells) return;
    const int hw = p.h * p.w;
    const int img = cell / hw, rem = cell - img * hw;
    const int y = rem / p.w, x = rem - y * p.w;
    const float *mk = p.mask + (long long)cell * 576 + lane;
    float m[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) m[k] = mk[k * 64];
    // the nine neighbours' coarse values: wave-uniform -> scalar loads, ALL issued here, branch-free (an outside
    // neighbour is read at a clamped address and weighted 0) -- as a chain of nine bounds-checked round trips
    // behind the softmax they were the kernel's critical path
    float nfx[9], nfy[9], nl0[9], nl1[9], nu[9];
    bool nok[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int ny = y + k / 3 - 1, nx = x + k % 3 - 1;
        nok[k] = ny >= 0 && ny < p.h && nx >= 0 && nx < p.w;
        const int cy = min(max(ny, 0), p.h - 1), cx = min(max(nx, 0), p.w - 1);
        const long long nc = ((long long)img * p.h + cy) * p.w + cx;
        const float2 f = reinterpret_cast<const float2 *>(p.flow_lr)[nc];
        const float *o = p.ou + nc * p.ld_ou;
        nfx[k] = f.x; nfy[k] = f.y; nl0[k] = o[0]; nl1[k] = o[1]; nu[k] = o[2];
    }
    float mx = m[0];
#pragma unroll
    for (int k = 1; k < 9; ++k) mx = fmaxf(mx, m[k]);
    float den = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { m[k] = expf(m[k] - mx); den += m[k]; }
    float fxv = 0.f, fyv = 0.f, l0 = 0.f, l1 = 0.f, u = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const float wk = nok[k] ? m[k] / den : 0.f;              // zeros outside (core/raft.py:88: unfold pads with 0)
        fxv += wk * (8.f * nfx[k]);
        fyv += wk * (8.f * nfy[k]);
        l0 += wk * nl0[k];
        l1 += wk * nl1[k];
        u += wk * nu[k];
    }
    const int Y0 = 8 * y + (lane >> 3) - p.pt, X0 = 8 * x + (lane & 7) - p.pl;   // unpadded coordinates
    if (Y0 < 0 || Y0 >= p.H0 || X0 < 0 || X0 >= p.W0) return;
    const long long plane = (long long)p.H0 * p.W0;
    const long long pix = (long long)Y0 * p.W0 + X0;
    // softmax over the two logits, channel 1
    const float lm = fmaxf(l0, l1);
    const float e0 = expf(l0 - lm), e1 = expf(l1 - lm);
    const float oc = e1 / (e0 + e1);
    const float sg = sqrtf(expf(u));
    if (p.flow != nullptr) {                       // planar outputs are optional when the packed form is asked for
        p.flow[(img * 2 + 0) * plane + pix] = fxv;
        p.flow[(img * 2 + 1) * plane + pix] = fyv;
        p.occl[img * plane + pix] = oc;
        p.sigma[img * plane + pix] = sg;
    }
    if (p.packed != nullptr)
        reinterpret_cast<float4 *>(p.packed)[img * plane + pix] = make_float4(fxv, fyv, oc, sg);
    // A non-finite result (an activation beyond the fp16 range of the split arithmetic reaches the outputs as NaN, by design)
    // must not enter a tracker's memory unnoticed: counted here, where every output of a refinement passes -- one ballot per
    // wave, an atomic only when something IS wrong; the host reads the counter when it synchronises anyway.
    if (p.nonfinite != nullptr) {
        // (sigma = sqrt(exp(u)) = +inf for u > ~88.7 is a value the reference produces and tolerates -- MFT/raft.py:62 applies no
        // clamp -- and the selection handles it; what is counted is what the reference could not have produced from finite
        // activations: a NaN anywhere, or an infinite flow / occlusion)
        const bool bad = !(isfinite(fxv) && isfinite(fyv) && isfinite(oc)) || sg != sg;
        const unsigned long long b = __ballot(bad);
        if (b != 0ull && bad && (unsigned)lane == (unsigned)__builtin_ctzll(b)) atomicAdd(p.nonfinite, (unsigned)__builtin_popcountll(b));
    }
}

int launch_convex_upsample(const float *flow_lr, const float *ou, int ld_ou, const float *mask, int P, int h,
                           int w, int pl, int pr, int pt, int pb, float *flow, float *occl, float *sigma,
                           float *packed, hipStream_t s, unsigned *nonfinite) {
    UpArgs a;
    a.nonfinite = nonfinite;
    a.flow_lr = flow_lr; a.ou = ou; a.ld_ou = ld_ou; a.mask = mask;
    a.P = P; a.h = h; a.w = w; a.pl = pl; a.pt = pt;
    a.H0 = 8 * h - pt - pb; a.W0 = 8 * w - pl - pr;
    a.flow = flow; a.occl = occl; a.sigma = sigma; a.packed = packed;
    a.cells = P * h * w;
    if (a.H0 <= 0 || a.W0 <= 0) return fail(MFTX_E_ARG, "convex_upsample: bad padding");
    if (packed != nullptr && !aligned16(packed)) return fail(MFTX_E_ALIGN, "convex_upsample: packed output must be 16-byte aligned");
    // algorithmic bytes: mask + coarse inputs read, 4 values per output pixel written -- once, in
    // whichever of the two layouts; writing both layouts is 16 more bytes per pixel that are not booked
    ProfScope prof(PC_UPSAMPLE, s, (double)P * h * w * (576 + 5) * 4 + (double)P * 4 * a.H0 * a.W0 * 4);
    hipLaunchKernelGGL(convex_upsample_kernel, dim3(cdiv(a.cells, UP_WAVES)), dim3(64 * UP_WAVES), 0, s, a);
    return check_launch("convex_upsample");
}

}  // namespace mftx
