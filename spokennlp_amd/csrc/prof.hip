// Host side of the launch timer (prof.h): event pool, per-class totals.  Exported through the C ABI as amdseg_prof_enable /
// amdseg_prof_reset / amdseg_prof_read (include/amdseg.h).
#include <vector>
#include "prof.h"
#include "common.h"
#include "amdseg_internal.h"

namespace {
struct Rec { int cls; double work; hipEvent_t e0, e1; };
constexpr size_t CAP = 65536;                       // profiled launches between two resets
bool g_on = false;
std::vector<Rec> g_recs;                            // [0, g_used) are live, the rest are pooled events of earlier rounds
size_t g_used = 0;
bool g_overflow = false;
}  // namespace

bool amdseg_prof_events(int cls, double work, hipEvent_t* start, hipEvent_t* stop) {
    if (!g_on) return false;
    if (g_used >= CAP) { g_overflow = true; return false; }
    if (g_used == g_recs.size()) {
        Rec r{};
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return false;
        g_recs.push_back(r);
    }
    Rec& r = g_recs[g_used++];
    r.cls = cls; r.work = work;
    *start = r.e0; *stop = r.e1;
    return true;
}

extern "C" {
int amdseg_prof_enable(int on) {
    const int prev = g_on ? 1 : 0;
    g_on = on != 0;
    return prev;
}
int amdseg_prof_reset(void) {
    hipError_t e = hipDeviceSynchronize();
    g_used = 0; g_overflow = false;
    return (int)e;
}
int amdseg_prof_read(int cls, double* total_us, double* total_work, long long* launches) {
    if (!total_us || !total_work || !launches) return AMDSEG_ERR_ARG;
    *total_us = 0; *total_work = 0; *launches = 0;
    if (g_used == 0) return 0;
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return (int)e;
    for (size_t i = 0; i < g_used; ++i) {
        const Rec& r = g_recs[i];
        if (r.cls != cls) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
        *total_us += (double)ms * 1000.0;
        *total_work += r.work;
        *launches += 1;
    }
    return g_overflow ? AMDSEG_ERR_SHAPE : 0;
}
}
