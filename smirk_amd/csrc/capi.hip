// Error strings / ABI version of libsmirk_hip.so.
#include "common.h"

extern "C" const char* smirk_strerror(int code) {
    switch (code) {
        case SMIRK_OK: return "ok";
        case SMIRK_ERR_BAD_ARG: return "bad argument (null pointer, size or alignment constraint violated)";
        case SMIRK_ERR_WORKSPACE: return "workspace too small (see smirk_*_workspace_bytes)";
        case SMIRK_ERR_LAUNCH: return "HIP kernel launch failed";
        case SMIRK_ERR_UNSUPPORTED: return "configuration not supported by the gfx950 kernels";
        default: return "unknown smirk error";
    }
}

extern "C" int smirk_abi_version(void) { return 11; }

// ---- split-fp16 range flag (common.h: smirk_range_audit8) -------------------------------------------------------------------------------
// ONE host-pinned, device-mapped word per process.  Kernels store a 1 into it (system-scope store, cold path) when a value they are about to split into the
// storage format would overflow the fp16 `hi` half; the host READS it with a plain load — no stream or device synchronisation — when the next forward of a
// module starts (or when the user asks, smirk_range_flag_peek after a synchronisation of their own).  Sticky until smirk_range_flag_clear().
#include <mutex>
namespace {
std::once_flag g_range_once;
unsigned* g_range_host = nullptr;
unsigned* g_range_dev = nullptr;
}  // namespace
unsigned* smirk_range_flag_device_ptr() {
    std::call_once(g_range_once, [] {
        void* h = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess || !h) { (void)hipGetLastError(); return; }
        *(volatile unsigned*)h = 0u;
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess || !d) { (void)hipGetLastError(); (void)hipHostFree(h); return; }
        g_range_host = (unsigned*)h;
        g_range_dev = (unsigned*)d;
    });
    return g_range_dev;
}
extern "C" unsigned smirk_range_flag_peek(void) { return g_range_host ? *(volatile unsigned*)g_range_host : 0u; }
extern "C" void smirk_range_flag_clear(void) { if (g_range_host) *(volatile unsigned*)g_range_host = 0u; }

// ---- launch profiler ---------------------------------------------------------------------------------------------------------------
#include <mutex>
#include <string.h>
#include <vector>

bool g_smirk_prof_on = false;
namespace {
struct ProfRec { char name[120]; double flop, bytes; hipEvent_t e0, e1; };
std::vector<ProfRec> g_recs;
std::mutex g_prof_mu;
thread_local char t_next_name[120];
thread_local double t_next_flop = 0.0, t_next_bytes = 0.0;
thread_local bool t_has_name = false;
thread_local long t_open = -1;
}  // namespace

void smirk_prof_next(const char* name, double flop, double bytes) {
    if (!g_smirk_prof_on) return;
    t_has_name = name != nullptr;
    if (name) { strncpy(t_next_name, name, sizeof(t_next_name) - 1); t_next_name[sizeof(t_next_name) - 1] = 0; }
    t_next_flop = flop; t_next_bytes = bytes;
}

void smirk_prof_begin(const char* name, hipStream_t st) {
    ProfRec r;
    const char* src = t_has_name ? t_next_name : name;
    // "(kernel<1, 2>)" -> "kernel<1,2>"
    size_t o = 0;
    for (const char* c = src; *c && o + 1 < sizeof(r.name); ++c) {
        if (*c == ' ' || ((*c == '(' ) && c == src)) continue;
        r.name[o++] = *c;
    }
    if (o > 0 && r.name[o - 1] == ')' && src[0] == '(') --o;
    r.name[o] = 0;
    r.flop = t_next_flop; r.bytes = t_next_bytes;
    t_has_name = false; t_next_flop = t_next_bytes = 0.0;
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) { t_open = -1; return; }
    (void)hipEventRecord(r.e0, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_recs.push_back(r);
    t_open = (long)g_recs.size() - 1;
}

void smirk_prof_end(hipStream_t st) {
    if (t_open < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if ((size_t)t_open < g_recs.size()) (void)hipEventRecord(g_recs[t_open].e1, st);
    t_open = -1;
}

extern "C" int smirk_profile_start(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_recs.clear();
    g_smirk_prof_on = true;
    return SMIRK_OK;
}

extern "C" int smirk_profile_stop(SmirkProfileRecord* out, int cap) {
    g_smirk_prof_on = false;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    const int n = (int)g_recs.size();
    for (int i = 0; i < n; ++i) {
        ProfRec& r = g_recs[i];
        float ms = 0.f;
        (void)hipEventSynchronize(r.e1);
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) ms = -1.f;
        if (out && i < cap) {
            memset(&out[i], 0, sizeof(out[i]));
            strncpy(out[i].kernel, r.name, sizeof(out[i].kernel) - 1);
            out[i].flop = r.flop; out[i].bytes = r.bytes; out[i].ms = ms;
        }
        (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
    }
    g_recs.clear();
    return n;
}
