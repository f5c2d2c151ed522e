/*
 * CPU ORACLE (plain C) for the kernels of the MFT hot path -- TEST
 * INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load the library built from this file.
 *
 * Scalar restatement of the reference algorithm (citations relative to
 * /root/reference), independent of torch so that it cross-checks
 * oracle/mft_oracle.py as well as the HIP kernels.  Pinned against the
 * reference-generated vectors in tests/golden/ by tests/test_oracle_golden.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* core/corr.py:53-69: V[i][j] = <f1[:,i], f2[:,j]> / sqrt(C); f* are [C][N] */
void orc_corr_volume(const float *f1, const float *f2, int C, int N, float *vol) {
    const float inv = 1.0f / sqrtf((float)C);
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            float acc = 0.f;
            for (int c = 0; c < C; ++c) acc += f1[(size_t)c * N + i] * f2[(size_t)c * N + j];
            vol[(size_t)i * N + j] = acc * inv;
        }
}

/* core/corr.py:26-28: avg_pool2d(2, stride 2) over [rows][h][w], floor sizes */
void orc_avg_pool2(const float *src, int rows, int h, int w, float *dst) {
    const int h2 = h / 2, w2 = w / 2;
    for (int r = 0; r < rows; ++r)
        for (int y = 0; y < h2; ++y)
            for (int x = 0; x < w2; ++x) {
                const float *p = src + ((size_t)r * h + 2 * y) * w + 2 * x;
                dst[((size_t)r * h2 + y) * w2 + x] = (((p[0] + p[1]) + p[w]) + p[w + 1]) * 0.25f;
            }
}

static float tap(const float *img, int H, int W, long y, long x) {
    return (y >= 0 && y < H && x >= 0 && x < W) ? img[(size_t)y * W + x] : 0.f;
}

/* F.grid_sample(bilinear, zeros, align_corners=True) at un-normalised (ix, iy) */
static float bilerp(const float *img, int H, int W, float ix, float iy) {
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const float wx = ix - fx0, wy = iy - fy0;
    const long x0 = (long)fmaxf(fminf(fx0, 1e9f), -1e9f), y0 = (long)fmaxf(fminf(fy0, 1e9f), -1e9f);
    return tap(img, H, W, y0, x0) * ((1.f - wx) * (1.f - wy)) + tap(img, H, W, y0, x0 + 1) * (wx * (1.f - wy)) +
           tap(img, H, W, y0 + 1, x0) * ((1.f - wx) * wy) + tap(img, H, W, y0 + 1, x0 + 1) * (wx * wy);
}

/* core/corr.py:30-51 + core/utils/utils.py:98-112.  lvl: [N][hl*wl]; coords
 * [2][N] (x then y); out [L*81][N], channel l*81 + a*9 + b at
 * (x/2^l + a - r, y/2^l + b - r) with the 2x/(W-1)-1 round trip. */
void orc_corr_lookup(const float *const *lvl, const int *hl, const int *wl, int L, int r, const float *coords,
                     int N, float *out) {
    const int d = 2 * r + 1;
    for (int i = 0; i < N; ++i)
        for (int l = 0; l < L; ++l) {
            const int H = hl[l], W = wl[l];
            const float cx = coords[i] / (float)(1 << l), cy = coords[(size_t)N + i] / (float)(1 << l);
            for (int a = 0; a < d; ++a)
                for (int b = 0; b < d; ++b) {
                    const float px = cx + (float)(a - r), py = cy + (float)(b - r);
                    const float gx = 2.f * px / (float)(W - 1) - 1.f, gy = 2.f * py / (float)(H - 1) - 1.f;
                    const float ix = ((gx + 1.f) / 2.f) * (float)(W - 1), iy = ((gy + 1.f) / 2.f) * (float)(H - 1);
                    out[((size_t)l * d * d + a * d + b) * N + i] = bilerp(lvl[l] + (size_t)i * H * W, H, W, ix, iy);
                }
        }
}

/* nn.Conv2d, zero "same" padding, NCHW, batch 1 (core/update.py) */
void orc_conv2d(const float *x, int Cin, int H, int W, const float *wgt, const float *bias, int Cout, int kh, int kw,
                float *out) {
    const int py = kh / 2, px = kw / 2;
    for (int co = 0; co < Cout; ++co)
        for (int y = 0; y < H; ++y)
            for (int xo = 0; xo < W; ++xo) {
                float acc = bias ? bias[co] : 0.f;
                for (int ci = 0; ci < Cin; ++ci)
                    for (int ky = 0; ky < kh; ++ky) {
                        const int yy = y + ky - py;
                        if (yy < 0 || yy >= H) continue;
                        for (int kx = 0; kx < kw; ++kx) {
                            const int xx = xo + kx - px;
                            if (xx < 0 || xx >= W) continue;
                            acc += x[((size_t)ci * H + yy) * W + xx] * wgt[(((size_t)co * Cin + ci) * kh + ky) * kw + kx];
                        }
                    }
                out[((size_t)co * H + y) * W + xo] = acc;
            }
}

/* core/raft.py:83-94: x [C][h][w], mask [576][h][w] -> out [C][8h][8w] */
void orc_convex_upsample(const float *x, int C, int h, int w, const float *mask, float mult, float *out) {
    const size_t hw = (size_t)h * w;
    for (int y = 0; y < h; ++y)
        for (int xx = 0; xx < w; ++xx)
            for (int sy = 0; sy < 8; ++sy)
                for (int sx = 0; sx < 8; ++sx) {
                    float m[9], mx = -INFINITY, den = 0.f;
                    for (int k = 0; k < 9; ++k) {
                        m[k] = mask[((size_t)k * 64 + sy * 8 + sx) * hw + (size_t)y * w + xx];
                        mx = fmaxf(mx, m[k]);
                    }
                    for (int k = 0; k < 9; ++k) { m[k] = expf(m[k] - mx); den += m[k]; }
                    for (int c = 0; c < C; ++c) {
                        float acc = 0.f;
                        for (int k = 0; k < 9; ++k) {
                            const int ny = y + k / 3 - 1, nx = xx + k % 3 - 1;
                            if (ny < 0 || ny >= h || nx < 0 || nx >= w) continue;
                            acc += (m[k] / den) * (mult * x[(size_t)c * hw + (size_t)ny * w + nx]);
                        }
                        out[((size_t)c * 8 * h + 8 * y + sy) * 8 * w + 8 * xx + sx] = acc;
                    }
                }
}

/* MFT/MFT.py:233-239 + MFT/results.py:87-136 + MFT/utils/interpolation.py:69-72 */
void orc_chain(const float *flowL, const float *occL, const float *sigL, const float *flowR, const float *occR,
               const float *sigR, int H, int W, float *flowO, float *occO, float *sigO) {
    const size_t plane = (size_t)H * W;
    const float sx = (float)(2.0 / (double)(W - 1)), sy = (float)(2.0 / (double)(H - 1));
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t p = (size_t)y * W + x;
            const float px = (float)x + flowL[p], py = (float)y + flowL[plane + p];
            const float ix = ((px * sx - 1.f) + 1.f) / 2.f * (float)(W - 1);
            const float iy = ((py * sy - 1.f) + 1.f) / 2.f * (float)(H - 1);
            flowO[p] = (px + bilerp(flowR, H, W, ix, iy)) - (float)x;
            flowO[plane + p] = (py + bilerp(flowR + plane, H, W, ix, iy)) - (float)y;
            occO[p] = fmaxf(occL[p], bilerp(occR, H, W, ix, iy));
            const float sr = bilerp(sigR, H, W, ix, iy);
            sigO[p] = sqrtf(sigL[p] * sigL[p] + sr * sr);
        }
}

/* MFT/MFT.py:112-143 + MFT/results.py:250-265; candidates ordered [inf,1,2,..] */
void orc_select(int K, const float *const *flow, const float *const *occ, const float *const *sig, float thr, int H,
                int W, float *flowO, float *occO, float *sigO, int8_t *chosen) {
    const size_t plane = (size_t)H * W;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t p = (size_t)y * W + x;
            int best = 0;
            float bs = 0.f;
            for (int k = 0; k < K; ++k) {
                const float s = (occ[k][p] > thr) ? -INFINITY : -sig[k][p];
                if (k == 0 || s > bs) { bs = s; best = k; }
            }
            const float fx = flow[best][p], fy = flow[best][plane + p];
            const float qx = (float)x + fx, qy = (float)y + fy;
            flowO[p] = fx;
            flowO[plane + p] = fy;
            occO[p] = (qx < 0.f || qy < 0.f || qx >= (float)W || qy >= (float)H) ? 1.f : occ[best][p];
            sigO[p] = sig[best][p];
            if (chosen) chosen[p] = (int8_t)best;
        }
}
