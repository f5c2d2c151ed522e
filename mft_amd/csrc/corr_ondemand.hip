// On-demand correlation lookup -- the memory-light alternative of the reference
// (AlternateCorrBlock + alt_cuda_corr, MFT/RAFT/core/corr.py:72-100,
// alt_cuda_corr/correlation_kernel.cu:18-119): no N x N volume is stored; the second feature map is
// average-pooled three times (a few MB per pair) and every iteration evaluates, per query cell and level, the
// 10 x 10 dot products <f1[cell], f2_l[tap]> / sqrt(C) its 9 x 9 bilinear window needs.  Pooling the features
// equals pooling the volume over the target dims (linearity), so the output is the materialised lookup's up to
// fp32 rounding -- same output layout as mftx_corr_lookup, cross-checked against it in the tests.
//
// Cost model: 400 feature rows of 1 KiB per cell and iteration, from L2 (the pooled maps are L2-resident): the
// kernel is bound by the load path (64 B/clk/CU), ~6.4 k cycles per cell -- 4-5 x slower than volume + lookup at
// 512 x 512 (0.34 vs 0.07 ms per iteration at 7 pairs), about even at 1080p, where the volume GEMM grows with
// N^2 (42 ms and 29.8 GB per 7-pair frame) and this grows with N.
//
// One wave per cell.  Lane = (g, s): four tap groups g x 16 channel slices s; lane s owns channels
// 4 s + 64 j + (0..3), j = 0..3, so the 16 lanes of a group read 256 contiguous bytes of a tap's row per load
// and a tap's row takes 4 loads.  Step i handles taps 4 i + g: 25 steps per level.  The 25 partial sums of a lane
// are reduced across its 16-lane group by small transposing butterflies (8 shuffles per 5 steps instead of 4 per
// tap) and go to the LDS window patch; the
// bilinear blend and the output are those of the materialised lookup (motion_front.h).
#include "common.h"
#include "profile.h"
#include "motion_front.h"

namespace mftx {

// features, pixel-major [P][h*w][C] -> [P][(h/2)*(w/2)][C], 2 x 2 means in ATen's order (F.avg_pool2d, corr.py:80-82)
__global__ __launch_bounds__(256) void fmap_pool_kernel(const float *__restrict__ in, float *__restrict__ out, int P,
                                                        int h, int w, int c4) {
    const int ho = h >> 1, wo = w >> 1;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)P * ho * wo * c4;
    if (i >= total) return;
    const int q = (int)(i % c4);
    const long long cell = i / c4;
    const int x = (int)(cell % wo), y = (int)((cell / wo) % ho), p = (int)(cell / ((long long)wo * ho));
    const f32x4lk *src = reinterpret_cast<const f32x4lk *>(in) + ((long long)p * h * w) * c4 + q;
    const f32x4lk a = src[((long long)(2 * y) * w + 2 * x) * c4], b = src[((long long)(2 * y) * w + 2 * x + 1) * c4];
    const f32x4lk c = src[((long long)(2 * y + 1) * w + 2 * x) * c4], d = src[((long long)(2 * y + 1) * w + 2 * x + 1) * c4];
    reinterpret_cast<f32x4lk *>(out)[i] = (((a + b) + c) + d) * 0.25f;
}

struct OnDemandArgs {
    const float *f1;            // [P][h*w][256]
    const float *f2[4];         // level l: [P][h_l*w_l][256]
    const float *coords;
    float *out;
    int ld_out, cells, n_per_img;
    int hl[4], wl[4];
    float scale;
};

__global__ __launch_bounds__(64 * LK_WAVES) void corr_ondemand_kernel(OnDemandArgs p) {
    __shared__ __attribute__((aligned(16))) float taps[LK_WAVES][4 * LK_LVL + 16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int g = lane >> 4, s = lane & 15;
    const int w_lvl = (lane >> 2) & 3, w_idx = lane & 3;
    int o_lvl[6], o_ab[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int o = lane + 64 * j;
        const int l = min(o / 81, 3);
        const int rem = o - l * 81;
        const int a = rem / 9, b = rem - a * 9;
        o_lvl[j] = l;
        o_ab[j] = l * LK_LVL + b * LK_ROW + a;
    }
    const int n_waves = gridDim.x * LK_WAVES;
    for (int cv = blockIdx.x * LK_WAVES + wv; cv < p.cells; cv += n_waves) {
        const int cell = __builtin_amdgcn_readfirstlane(cv);
        const int pair = cell / p.n_per_img;
        f32x4lk f1v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            f1v[j] = *reinterpret_cast<const f32x4lk *>(p.f1 + (long long)cell * 256 + 64 * j + 4 * s);
        const float2 c = reinterpret_cast<const float2 *>(p.coords)[cell];
        float *tp = taps[wv];
        float my_fx = 0.f, my_fy = 0.f;
#pragma unroll 1
        for (int l = 0; l < 4; ++l) {
            const float sx = c.x / (float)(1 << l), sy = c.y / (float)(1 << l);
            const float flx = floorf(sx), fly = floorf(sy);
            if (w_lvl == l) { my_fx = sx - flx; my_fy = sy - fly; }
            const int x0 = (int)fminf(fmaxf(flx, -1.0e6f), 1.0e6f) - 4;
            const int y0 = (int)fminf(fmaxf(fly, -1.0e6f), 1.0e6f) - 4;
            const unsigned H = p.hl[l], W = p.wl[l];
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float *>(p.f2[l] + (long long)pair * H * W * 256), 0, H * W * 1024u, 0x00020000);
            // 25 steps in 5 rounds of 5 (a real loop: at most 20 row loads / 80 registers in flight); a round's 5
            // partial sums are reduced across the 16 lanes of the group by a small transposing butterfly
            // (8 slots: 4 + 2 + 1 exchanges, then one plain step), leaving slot (s >> 1) on lane s
#pragma unroll 1
            for (int i5 = 0; i5 < 5; ++i5) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = 0.f;
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const int t = 4 * (5 * i5 + k) + g;           // this group's tap at this step
                    const int tr = t / 10, tc = t - tr * 10;
                    const unsigned yy = (unsigned)(y0 + tr), xx = (unsigned)(x0 + tc);
                    const bool ok = (yy < H) & (xx < W);            // outside the level: zeros (grid_sample's zero padding)
                    const unsigned off = ok ? ((yy * W + xx) * 256u + 4u * (unsigned)s) * 4u : 0x80000000u;
                    float acc = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4lk r = __builtin_bit_cast(f32x4lk, __builtin_amdgcn_raw_buffer_load_b128(rs, off + 256u * j, 0, 0));
                        acc += f1v[j].x * r.x + f1v[j].y * r.y + f1v[j].z * r.z + f1v[j].w * r.w;
                    }
                    v[k] = acc;
                }
                int cnt = 8;
#pragma unroll
                for (int off = 8; off >= 2; off >>= 1) {
                    const bool upper = (lane & off) != 0;
                    const int half = cnt / 2;
#pragma unroll
                    for (int i = 0; i < half; ++i) {
                        const float send = upper ? v[i] : v[i + half];
                        const float keep = upper ? v[i + half] : v[i];
                        v[i] = keep + __shfl_xor(send, off);
                    }
                    cnt = half;
                }
                v[0] += __shfl_xor(v[0], 1);
                const int slot = s >> 1;
                if ((s & 1) == 0 && slot < 5) {
                    const int t = 4 * (5 * i5 + slot) + g;
                    const int tr = t / 10, tc = t - tr * 10;
                    tp[l * LK_LVL + tr * LK_ROW + tc] = v[0] * p.scale;
                }
            }
        }
        if (lane < 16) {
            const float ax = (w_idx & 1) ? my_fx : 1.f - my_fx;
            const float ay = (w_idx & 2) ? my_fy : 1.f - my_fy;
            tp[4 * LK_LVL + lane] = ax * ay;
        }
        // (LDS operations of one wave complete in issue order: no barrier needed to read other lanes' writes)
        float *dst = p.out + (long long)cell * p.ld_out;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int o = lane + 64 * j;
            if (j < 5 || o < 324) {
                const float4 wq = *reinterpret_cast<const float4 *>(tp + 4 * LK_LVL + o_lvl[j] * 4);
                const float *t4 = tp + o_ab[j];
                dst[o] = t4[0] * wq.x + t4[1] * wq.y + t4[LK_ROW] * wq.z + t4[LK_ROW + 1] * wq.w;
            }
        }
    }
}

int launch_fmap_pyramid(const float *f2, int P, int C, int h, int w, float *const lvl[3], hipStream_t s) {
    const float *src = f2;
    int hh = h, ww = w;
    for (int l = 0; l < 3; ++l) {
        const long long total = (long long)P * (hh >> 1) * (ww >> 1) * (C / 4);
        ProfScope prof(PC_CORR_POOL, s, 4.0 * P * C * ((double)hh * ww + (double)(hh >> 1) * (ww >> 1)));
        hipLaunchKernelGGL(fmap_pool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, lvl[l], P, hh, ww, C / 4);
        if (int e = check_launch("fmap_pool")) return e;
        src = lvl[l]; hh >>= 1; ww >>= 1;
    }
    return 0;
}

int launch_corr_ondemand(const float *f1, const float *const f2lvl[4], const float *coords, int P, int h, int w,
                         float *out, int ld_out, hipStream_t s) {
    OnDemandArgs a;
    a.f1 = f1; a.coords = coords; a.out = out; a.ld_out = ld_out;
    a.cells = P * h * w; a.n_per_img = h * w;
    for (int l = 0; l < 4; ++l) { a.f2[l] = f2lvl[l]; a.hl[l] = h >> l; a.wl[l] = w >> l; }
    if ((long long)h * w * 1024 > 0x7fffffffLL) return fail(MFTX_E_ARG, "corr_lookup_ondemand: feature map exceeds 2 GiB");
    a.scale = 1.0f / sqrtf(256.f);
    const int blocks = cdiv(a.cells, LK_WAVES);
    // booked like the materialised lookup (same algorithmic result): taps + coords + outputs
    ProfScope prof(PC_LOOKUP, s, (double)a.cells * (4 * 100 * 4 + 8 + 324 * 4));
    hipLaunchKernelGGL(corr_ondemand_kernel, dim3(blocks), dim3(64 * LK_WAVES), 0, s, a);
    return check_launch("corr_lookup_ondemand");
}

}  // namespace mftx
