// Version + thread-local error string of libmftx.
#include "common.h"
#include <cstring>

namespace mftx {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
}  // namespace mftx

extern "C" int mftx_version(void) { return MFTX_VERSION; }
extern "C" const char *mftx_last_error_string(void) { return mftx::g_err; }
