// The explicit library context (include/amdseg.h, ABI 13: amdseg_ctx_*): EVERYTHING libamdseg remembers between two calls lives in an amdseg_ctx
// the caller created -- the CU budget of the tile rules, the small-tile test hook, and the launch timer (event pool, per-class totals).  A call finds
// its context in amdseg_bert_cfg.ctx (the composite layer calls) or, for entry points without a cfg, in the context the calling thread bound with
// amdseg_ctx_bind(); with neither the built-in defaults apply (every CU, the tile rules' own choice, no timing).  The only thread-local word is that
// binding -- a pointer to a caller-owned object, not state of the library.
// Launch timer: a profiled launch goes through hipExtLaunchKernelGGL with a start and a stop event (prof.h).
#include <vector>
#include "prof.h"
#include "common.h"
#include "amdseg_internal.h"

namespace {
constexpr size_t CAP = 65536;                       // profiled launches between two resets
thread_local amdseg_ctx* t_ctx = nullptr;           // the context of the call in progress (cfg.ctx) or the one this thread bound
}  // namespace

amdseg_ctx* amdseg_current_ctx() { return t_ctx; }
AmdsegCtxScope::AmdsegCtxScope(amdseg_ctx* c) : prev(t_ctx), active(c != nullptr) { if (active) t_ctx = c; }
AmdsegCtxScope::~AmdsegCtxScope() { if (active) t_ctx = prev; }

bool amdseg_prof_events(int cls, double work, hipEvent_t* start, hipEvent_t* stop) {
    amdseg_ctx* c = t_ctx;
    if (!c || !c->prof_on) return false;
    if (c->used >= CAP) { c->overflow = true; return false; }
    if (c->used == c->recs.size()) {
        AmdsegProfRec r{};
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return false;
        c->recs.push_back(r);
    }
    AmdsegProfRec& r = c->recs[c->used++];
    r.cls = cls; r.work = work;
    *start = r.e0; *stop = r.e1;
    return true;
}

extern "C" {
int amdseg_ctx_create(amdseg_ctx** out) {
    if (!out) return AMDSEG_ERR_ARG;
    *out = new (std::nothrow) amdseg_ctx();
    return *out ? AMDSEG_OK : AMDSEG_ERR_ARG;
}
int amdseg_ctx_destroy(amdseg_ctx* c) {
    if (!c) return AMDSEG_OK;
    if (t_ctx == c) t_ctx = nullptr;
    for (AmdsegProfRec& r : c->recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    delete c;
    return AMDSEG_OK;
}
int amdseg_ctx_bind(amdseg_ctx* c) { t_ctx = c; return AMDSEG_OK; }
int amdseg_ctx_set_cu_budget(amdseg_ctx* c, int cus) {
    if (!c) return AMDSEG_ERR_ARG;
    const int prev = c->cu_budget;
    c->cu_budget = cus > 0 ? cus : 0;
    return prev;
}
int amdseg_ctx_cu_budget(const amdseg_ctx* c) { return c ? c->cu_budget : 0; }
int amdseg_ctx_force_small_tile(amdseg_ctx* c, int v) {
    if (!c) return AMDSEG_ERR_ARG;
    const int prev = c->force_small_tile;
    c->force_small_tile = v;
    return prev;
}
int amdseg_ctx_prof_enable(amdseg_ctx* c, int on) {
    if (!c) return AMDSEG_ERR_ARG;
    const int prev = c->prof_on ? 1 : 0;
    c->prof_on = on != 0;
    return prev;
}
int amdseg_ctx_prof_reset(amdseg_ctx* c) {
    if (!c) return AMDSEG_ERR_ARG;
    hipError_t e = hipDeviceSynchronize();
    c->used = 0; c->overflow = false;
    return (int)e;
}
int amdseg_ctx_prof_read(amdseg_ctx* c, int cls, double* total_us, double* total_work, long long* launches) {
    if (!c || !total_us || !total_work || !launches) return AMDSEG_ERR_ARG;
    *total_us = 0; *total_work = 0; *launches = 0;
    if (c->used == 0) return 0;
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return (int)e;
    for (size_t i = 0; i < c->used; ++i) {
        const AmdsegProfRec& r = c->recs[i];
        if (r.cls != cls) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) continue;
        *total_us += (double)ms * 1000.0;
        *total_work += r.work;
        *launches += 1;
    }
    return c->overflow ? AMDSEG_ERR_SHAPE : 0;
}
}
