// Optional per-kernel timing with HIP events on the launch stream (bench.py's
// roofline leg).  Off by default: zero cost in the product path.
#pragma once
#include <hip/hip_runtime.h>

namespace mftx {

enum ProfCat { PC_CORR_VOLUME, PC_CORR_POOL, PC_LOOKUP, PC_CONV_GEMM, PC_CONVF1, PC_GLUE, PC_UPSAMPLE, PC_CHAIN,
               PC_CONV_SMALL, PC_ENC_NORM, PC_LOOKUP_FUSED, PC_FLOW_FUSED, PC_ENC_GEMM, PC_GRU_FUSED, PC_COUNT };

// the category a conv GEMM launched inside this scope is booked under instead of PC_CONV_GEMM (the encoders' layers: their own
// line in bench.py's `kernels`); profiler state only
struct ProfConvCat {
    static int &current() { static thread_local int c = -1; return c; }
    int saved;
    explicit ProfConvCat(ProfCat c) : saved(current()) { current() = (int)c; }
    ~ProfConvCat() { current() = saved; }
};

bool prof_enabled();
// bracket one launch: begin() records an event, end() records another and
// books `work` (algorithmic flops or bytes) for the category
void prof_begin(ProfCat c, hipStream_t s);
void prof_end(ProfCat c, hipStream_t s, double work);

struct ProfScope {
    ProfCat c; hipStream_t s; double work; bool on;
    ProfScope(ProfCat c_, hipStream_t s_, double work_) : c(c_), s(s_), work(work_), on(prof_enabled()) {
        if (on && c == PC_CONV_GEMM && ProfConvCat::current() >= 0) c = (ProfCat)ProfConvCat::current();
        if (on) prof_begin(c, s);
    }
    ~ProfScope() { if (on) prof_end(c, s, work); }
};

}  // namespace mftx
