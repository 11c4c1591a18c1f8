// Error plumbing and version for the C ABI (include/ivosw.h).
#include <stdarg.h>
#include <stdlib.h>

#include "common.h"

namespace ivosw {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

DeviceScope::DeviceScope(const void* p) {
    hipPointerAttribute_t at{};
    if (!p || hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();   // not a HIP allocation: clear the sticky error, the caller reports IVOSW_ERR_ARG
        return;
    }
    if (at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeManaged) return;
    dev = at.device;
    if (hipGetDevice(&prev) != hipSuccess) return;
    if (prev != dev && hipSetDevice(dev) != hipSuccess) return;
    ok = true;
}

DeviceScope::~DeviceScope() {
    if (ok && prev >= 0 && prev != dev) (void)hipSetDevice(prev);
}
}  // namespace ivosw

namespace ivosw {
// Tunables: small fixed table, looked up by name.  Not thread-safe against concurrent ivosw_tune_set (tuning/test hook).
struct Tunable { char key[32]; int value; };
static Tunable g_tun[32];
static int g_ntun = 0;

static Tunable* tune_find(const char* key) {
    for (int i = 0; i < g_ntun; ++i)
        if (strcmp(g_tun[i].key, key) == 0) return &g_tun[i];
    return nullptr;
}

int tune_get(const char* key, int dflt) {
    if (Tunable* t = tune_find(key)) return t->value;
    char env[64];
    snprintf(env, sizeof(env), "IVOSW_TUNE_%s", key);
    const char* e = getenv(env);
    return e ? atoi(e) : dflt;
}
}  // namespace ivosw

extern "C" int ivosw_tune_set(const char* key, int value) {
    using namespace ivosw;
    IVOSW_REQUIRE(key && strlen(key) < sizeof(g_tun[0].key), "bad key");
    Tunable* t = tune_find(key);
    if (!t) {
        IVOSW_REQUIRE(g_ntun < 32, "tunable table full");
        t = &g_tun[g_ntun++];
        strcpy(t->key, key);
    }
    t->value = value;
    return IVOSW_OK;
}

extern "C" const char* ivosw_last_error(void) { return ivosw::g_err; }
extern "C" int ivosw_version(void) { return 100; }
