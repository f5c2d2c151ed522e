// Version, thread-local error string and the optional event profiler of libmftx.
#include "common.h"
#include "profile.h"
#include <cstring>
#include <vector>

namespace mftx {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// ---- profiler -------------------------------------------------------------
struct ProfRec { hipEvent_t a, b; int cat; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static hipEvent_t g_open[PC_COUNT];
static double g_work[PC_COUNT];
static long long g_count[PC_COUNT];

bool prof_enabled() { return g_prof_on; }

static hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

void prof_begin(ProfCat c, hipStream_t s) {
    g_open[c] = get_event();
    (void)hipEventRecord(g_open[c], s);
}

void prof_end(ProfCat c, hipStream_t s, double work) {
    hipEvent_t b = get_event();
    (void)hipEventRecord(b, s);
    g_recs.push_back(ProfRec{g_open[c], b, (int)c});
    g_work[c] += work;
    g_count[c] += 1;
}
}  // namespace mftx

using namespace mftx;

extern "C" int mftx_version(void) { return MFTX_VERSION; }
extern "C" const char *mftx_last_error_string(void) { return mftx::g_err; }

extern "C" int mftx_profile_begin(void) {
    for (auto &r : g_recs) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
    g_recs.clear();
    for (int i = 0; i < PC_COUNT; ++i) { g_work[i] = 0; g_count[i] = 0; }
    g_prof_on = true;
    return 0;
}

extern "C" int mftx_profile_end(double *ms, double *work, long long *count, int n) {
    g_prof_on = false;
    if (!ms || !work || !count || n != PC_COUNT) return fail(MFTX_E_ARG, "profile_end: need %d slots", (int)PC_COUNT);
    for (int i = 0; i < PC_COUNT; ++i) { ms[i] = 0; work[i] = g_work[i]; count[i] = g_count[i]; }
    for (auto &r : g_recs) {
        hipError_t e = hipEventSynchronize(r.b);
        if (e != hipSuccess) return fail((int)e, "profile_end: %s", hipGetErrorString(e));
        float t = 0.f;
        (void)hipEventElapsedTime(&t, r.a, r.b);
        ms[r.cat] += t;
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    g_recs.clear();
    return 0;
}
