// Shared host-side helpers for libmftx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/mftx.h"

namespace mftx {

void set_error(const char *fmt, ...);

inline int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    set_error("%s", buf);
    return code;
}

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int round_up(int a, int b) { return cdiv(a, b) * b; }

// ---- kernel launchers shared between the per-op exports and the RAFT engine
int launch_conv(const mftx_conv_desc &d, hipStream_t s);
int launch_conv_pair(const mftx_conv_desc &a, const mftx_conv_desc &b, hipStream_t s);   // two independent ReLU convs, one launch
// conv whose epilogue is a GRU gate (see conv_gemm.hip)
struct GruEpilogue {
    int mode;          // 1: z|r gates  2: candidate + blend
    float *hx;         // [M][ld_hx], hidden state in channels [0,128)
    int ld_hx;
    float *z;          // [M][128]
    float *rh;         // [M][128]
};
int launch_conv_gru(const mftx_conv_desc &d, const GruEpilogue &g, hipStream_t s);
bool conv_small_applicable(const mftx_conv_desc &d);
int launch_conv_small(const mftx_conv_desc &d, hipStream_t s, float *accum = nullptr, int ld_accum = 0);
int launch_corr_volume(const float *f1, const float *f2, int P, int C, int N, float *lvl0, hipStream_t s);
int launch_corr_pool(const float *lvl0, int rows, int h, int w, float *lvl1, float *lvl2, float *lvl3, hipStream_t s);
int launch_corr_lookup(const float *const lvl[4], const float *coords, int P, int h, int w,
                       float *out, int ld_out, hipStream_t s);
int launch_convex_upsample(const float *flow_lr, const float *ou, int ld_ou, const float *mask,
                           int P, int h, int w, int pl, int pr, int pt, int pb,
                           float *flow, float *occl, float *sigma, hipStream_t s);

}  // namespace mftx
