// Error plumbing and version for the C ABI (include/ivosw.h).
#include <stdarg.h>

#include "common.h"

namespace ivosw {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace ivosw

extern "C" const char* ivosw_last_error(void) { return ivosw::g_err; }
extern "C" int ivosw_version(void) { return 100; }
