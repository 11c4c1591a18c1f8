// Error plumbing and version for the C ABI (include/ivosw.h).
#include <stdarg.h>
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <vector>

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
// Tunables: fixed table, looked up by name.  Lookups, first-use insertion, tune_set and the one-time environment read are safe from
// several host threads: a SET value and the cached environment default live in separate atomic fields, so a lookup
// that is reading the environment can never overwrite a concurrent tune_set; the flags are published with release / acquire.
// The sources read ~60 distinct keys (grep tune_get; tests/test_cabi.py sets every one of them plus 40 more); ONE entry per key holds the
// set value and the cached environment default (IVOSW_TUNE_<KEY> is read ONCE per key and process: a forward pass asks for ~40 keys).
struct Tunable {
    char key[32];
    std::atomic<int> value;          // ivosw_tune_set
    std::atomic<int> env_value;      // IVOSW_TUNE_<KEY>, read once
    std::atomic<bool> set;
    std::atomic<int> env_state;      // 0 not read yet, 1 absent, 2 present
};
constexpr int kMaxTunables = 160;
static_assert(kMaxTunables >= 2 * 64, "the tunables table must hold every key the sources read (~60, one entry each), with room to grow");
static Tunable g_tun[kMaxTunables];
static std::atomic<int> g_ntun{0};
static std::mutex g_tun_mu;

// Entries are append-only: lookups scan without a lock (the count is read with acquire, an entry is complete before it is published);
// appending — a key's first tune_get / tune_set, from whichever host thread — takes the mutex and re-checks.
static Tunable* tune_find(const char* key, bool create) {
    int n = g_ntun.load(std::memory_order_acquire);
    for (int i = 0; i < n; ++i)
        if (strcmp(g_tun[i].key, key) == 0) return &g_tun[i];
    if (!create || strlen(key) >= sizeof(g_tun[0].key)) return nullptr;
    std::lock_guard<std::mutex> lock(g_tun_mu);
    const int m = g_ntun.load(std::memory_order_acquire);
    for (int i = n; i < m; ++i)
        if (strcmp(g_tun[i].key, key) == 0) return &g_tun[i];
    if (m >= kMaxTunables) return nullptr;
    Tunable* t = &g_tun[m];
    strcpy(t->key, key);
    t->value.store(0, std::memory_order_relaxed);
    t->env_value.store(0, std::memory_order_relaxed);
    t->set.store(false, std::memory_order_relaxed);
    t->env_state.store(0, std::memory_order_relaxed);
    g_ntun.store(m + 1, std::memory_order_release);
    return t;
}

int tune_get(const char* key, int dflt) {
    Tunable* t = tune_find(key, true);
    if (!t) {                                        // table full / key too long: uncached environment lookup, still correct
        char env[96];
        snprintf(env, sizeof(env), "IVOSW_TUNE_%s", key);
        const char* e = getenv(env);
        return e ? atoi(e) : dflt;
    }
    if (t->set.load(std::memory_order_acquire)) return t->value.load(std::memory_order_relaxed);
    int st = t->env_state.load(std::memory_order_acquire);
    if (st == 0) {                                   // several threads may get here together: they all compute the same two values
        char env[64];
        snprintf(env, sizeof(env), "IVOSW_TUNE_%s", key);
        const char* e = getenv(env);
        if (e) t->env_value.store(atoi(e), std::memory_order_relaxed);
        st = e ? 2 : 1;
        t->env_state.store(st, std::memory_order_release);
    }
    return st == 2 ? t->env_value.load(std::memory_order_relaxed) : dflt;
}
}  // namespace ivosw

extern "C" int ivosw_tune_set(const char* key, int value) {
    using namespace ivosw;
    IVOSW_REQUIRE(key && strlen(key) < sizeof(g_tun[0].key), "bad key");
    Tunable* t = tune_find(key, true);
    IVOSW_REQUIRE(t != nullptr, "tunable table full");
    t->value.store(value, std::memory_order_relaxed);
    t->set.store(true, std::memory_order_release);
    return IVOSW_OK;
}

extern "C" const char* ivosw_last_error(void) { return ivosw::g_err; }
extern "C" int ivosw_version(void) { return 100; }
extern "C" int ivosw_ablation_build(void) { return IVOSW_ABLATION; }


// ---------------------------------------------------------------- HIP graph capture of a launch sequence
// A DQN step is ~40 dependent launches of 5-30 us whose host enqueue (170-250 us) is as long as the step: capturing the
// calls once and replaying the graph costs one hipGraphLaunch per step.  The calls between begin and end are the ordinary
// entry points of this library on `stream` (they fork / join the library's helper stream with events, which joins the
// capture); nothing is allocated on the device, the graph is a host object owned by the handle.
namespace {
struct GraphHandle {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int kernel_nodes = 0, nodes = 0;
};
}  // namespace

extern "C" int ivosw_graph_begin(ivosw_stream_t stream) {
    IVOSW_REQUIRE(stream != nullptr, "capture needs an explicit (non-null) stream");
    hipError_t e = hipStreamBeginCapture(ivosw::as_stream(stream), hipStreamCaptureModeRelaxed);
    if (e != hipSuccess) {
        ivosw::set_error("ivosw_graph_begin: %s", hipGetErrorString(e));
        return IVOSW_ERR_LAUNCH;
    }
    return IVOSW_OK;
}

extern "C" int ivosw_graph_end(ivosw_stream_t stream, ivosw_graph_t* out, int* kernel_nodes) {
    IVOSW_REQUIRE(stream != nullptr && out, "null pointer");
    GraphHandle* h = new GraphHandle();
    hipError_t e = hipStreamEndCapture(ivosw::as_stream(stream), &h->graph);
    if (e == hipSuccess) e = hipGraphInstantiate(&h->exec, h->graph, nullptr, nullptr, 0);
    if (e == hipSuccess) {
        size_t n = 0;
        if (hipGraphGetNodes(h->graph, nullptr, &n) == hipSuccess && n > 0) {
            std::vector<hipGraphNode_t> nodes(n);
            if (hipGraphGetNodes(h->graph, nodes.data(), &n) == hipSuccess) {
                h->nodes = (int)n;
                for (size_t i = 0; i < n; ++i) {
                    hipGraphNodeType t;
                    if (hipGraphNodeGetType(nodes[i], &t) == hipSuccess && t == hipGraphNodeTypeKernel) ++h->kernel_nodes;
                }
            }
        }
    }
    if (e != hipSuccess) {
        ivosw::set_error("ivosw_graph_end: %s", hipGetErrorString(e));
        if (h->exec) (void)hipGraphExecDestroy(h->exec);
        if (h->graph) (void)hipGraphDestroy(h->graph);
        delete h;
        (void)hipGetLastError();
        return IVOSW_ERR_LAUNCH;
    }
    if (kernel_nodes) *kernel_nodes = h->kernel_nodes;
    *out = h;
    return IVOSW_OK;
}

extern "C" int ivosw_graph_launch(ivosw_graph_t g, ivosw_stream_t stream) {
    IVOSW_REQUIRE(g, "null graph");
    hipError_t e = hipGraphLaunch(static_cast<GraphHandle*>(g)->exec, ivosw::as_stream(stream));
    if (e != hipSuccess) {
        ivosw::set_error("ivosw_graph_launch: %s", hipGetErrorString(e));
        return IVOSW_ERR_LAUNCH;
    }
    return IVOSW_OK;
}

extern "C" int ivosw_graph_destroy(ivosw_graph_t g) {
    if (!g) return IVOSW_OK;
    GraphHandle* h = static_cast<GraphHandle*>(g);
    if (h->exec) (void)hipGraphExecDestroy(h->exec);
    if (h->graph) (void)hipGraphDestroy(h->graph);
    delete h;
    return IVOSW_OK;
}
