// mg_runtime.cpp -- error strings, launch counters and the optional per-kernel
// HIP-event timing of libmontage_hip.so.  Host code only.
#include "mg_common.h"
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";
static std::atomic<uint64_t> g_launches[MG_K_COUNT];

void mg_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int mg_check_launch(int kernel_id, const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        mg_set_error("%s: kernel launch failed: %s", what, hipGetErrorString(e));
        return MG_E_LAUNCH;
    }
    if (kernel_id >= 0 && kernel_id < MG_K_COUNT) g_launches[kernel_id].fetch_add(1, std::memory_order_relaxed);
    return MG_OK;
}

// ---- profiling --------------------------------------------------------------
struct ProfRec { int kid; hipEvent_t a, b; double bytes; };
static std::atomic<int> g_prof_on{0};
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_free_events;

static hipEvent_t take_event()
{
    if (!g_free_events.empty()) { hipEvent_t e = g_free_events.back(); g_free_events.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

MgLaunchScope::MgLaunchScope(int kernel_id, hipStream_t s, double algorithmic_bytes)
    : kid(kernel_id), stream(s), bytes(algorithmic_bytes), rec(nullptr)
{
    if (!g_prof_on.load(std::memory_order_relaxed)) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    hipEvent_t a = take_event(), b = take_event();
    if (!a || !b) return;
    (void)hipEventRecord(a, stream);
    ProfRec* r = new ProfRec{kid, a, b, bytes};
    rec = r;
}

MgLaunchScope::~MgLaunchScope()
{
    if (!rec) return;
    ProfRec* r = (ProfRec*)rec;
    (void)hipEventRecord(r->b, stream);
    r->kid = kid;                        // the launcher may have re-labelled the scope (e.g. tiled -> stream)
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_recs.push_back(*r);
    delete r;
}

extern "C" {

int mg_abi_version(void) { return MG_ABI_VERSION; }
const char* mg_last_error(void) { return g_err; }
uint64_t mg_launch_count(int kernel_id)
{
    return (kernel_id >= 0 && kernel_id < MG_K_COUNT) ? g_launches[kernel_id].load() : 0;
}

int mg_prof_enable(int on) { g_prof_on.store(on ? 1 : 0); return MG_OK; }

int mg_prof_reset(void)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_recs) { g_free_events.push_back(r.a); g_free_events.push_back(r.b); }
    g_recs.clear();
    return MG_OK;
}

int mg_prof_collect(int kernel_id, uint64_t* launches, double* ms, double* bytes)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    uint64_t n = 0; double t = 0, by = 0;
    for (auto& r : g_recs) {
        if (r.kid != kernel_id) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) { mg_set_error("mg_prof_collect: hipEventSynchronize failed"); return MG_E_LAUNCH; }
        float f = 0;
        if (hipEventElapsedTime(&f, r.a, r.b) != hipSuccess) { mg_set_error("mg_prof_collect: hipEventElapsedTime failed"); return MG_E_LAUNCH; }
        n++; t += f; by += r.bytes;
    }
    if (launches) *launches = n;
    if (ms) *ms = t;
    if (bytes) *bytes = by;
    return MG_OK;
}

} // extern "C"
