// The multi-scale 9x9 correlation lookup (HBM-bound).  The pyramid itself -- volume and the three pooled
// levels -- is written by the volume GEMM's epilogue (csrc/conv_gemm.hip, layout in common.h).
#include "common.h"
#include "profile.h"
#include "motion_front.h"
#include <cstdlib>

namespace mftx {

// ---------------------------------------------------------------------------
// Lookup (core/corr.py:30-51, core/utils/utils.py:98-112).
//
// One wave per query cell.  For each of the 4 levels the 81 window samples
// share one fractional offset, so the cell needs only the 10x10 integer taps
// around floor(c / 2^l) - 4: the wave pulls those 400 taps (zero outside the
// level, matching grid_sample's zero padding tap by tap) into its private LDS
// slab, then every lane blends 4 neighbours for its output channels and writes
// them pixel-major, 256 contiguous bytes per store.  Algorithmic traffic per
// cell: 4*100*4 B read + 8 B coords + 324*4 B written.  Real traffic is set by the
// 128-byte line: levels 0 and 1 are stored in 8 x 4-float blocks (common.h), which
// a 10 x 10 window cuts in (1 + 9/8)(1 + 9/4) = 6.9 lines on average -- the floor
// for 32-float lines -- against 12.8 / 10 in a row-major level; with ~6 lines of
// level 2 and 2 of level 3 that is ~2.8 KB read per cell for 1.6 KB of unique taps.
//
// Coordinates: the reference normalises to [-1,1] and grid_sample maps back;
// that round trip moves a coordinate by O(1e-6) px, and zero-padded bilinear
// sampling is continuous in the coordinate, so sampling directly at
// c/2^l + (a-4) agrees to O(1e-6 * |grad V|).
// ---------------------------------------------------------------------------
int launch_corr_lookup(const float *const lvl[4], const float *coords, int P, int h, int w, float *out,
                       int ld_out, hipStream_t s) {
    const LookupArgs a = make_lookup_args(lvl, coords, P, h, w, out, ld_out);
    static const int cpw = tune_env("MFTX_LOOKUP_CPW", 2);
    // 4 levels x 10x10 unique taps read + coords + 324 outputs written, per cell
    ProfScope prof(PC_LOOKUP, s, lookup_bytes(a));
    if (cpw == 1)
        hipLaunchKernelGGL(corr_lookup_kernel<1>, dim3(cdiv(a.cells, LK_WAVES)), dim3(64 * LK_WAVES), 0, s, a);
    else if (cpw == 4)
        hipLaunchKernelGGL(corr_lookup_kernel<4>, dim3(cdiv(cdiv(a.cells, 4), LK_WAVES)), dim3(64 * LK_WAVES), 0, s, a);
    else
        hipLaunchKernelGGL(corr_lookup_kernel<2>, dim3(cdiv(cdiv(a.cells, 2), LK_WAVES)), dim3(64 * LK_WAVES), 0, s, a);
    return check_launch("corr_lookup");
}

}  // namespace mftx
