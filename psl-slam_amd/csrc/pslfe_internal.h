// Internal declarations of libpslfe (host side). Product code.
#ifndef PSLFE_INTERNAL_H
#define PSLFE_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/pslfe.h"

void pslfe_set_error(const char* fmt, ...);

#define PSL_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            pslfe_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_));  \
            return PSLFE_E_HIP;                                                                    \
        }                                                                                          \
    } while (0)

#define PSL_REQUIRE(cond, code, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            pslfe_set_error(__VA_ARGS__); \
            return (code);                \
        }                                 \
    } while (0)

struct StageTimer {
    double ms = 0;
    int launches = 0;
};

struct pslfe_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;  // stream in use (own or external)
    hipStream_t aux_stream = nullptr;              // second stream for independent kernels of one call (fork/join by events)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool profile = false;
    std::string profile_only;  // non-empty: only this stage is timed (two events per step instead of two per stage)
    std::map<std::string, StageTimer> stages;
    // event pool for profile mode: (start, stop, stage) triples resolved at synchronize
    struct Pending { hipEvent_t a, b; std::string stage; };
    std::vector<Pending> pending;
    int cu_count = 0;
    // Scratch arena of the host-pointer entry points (small per-call device buffers).  hipMalloc / hipFree per call cost ~10 us each and
    // hipFree waits for EVERY stream of the device - with a look-ahead extraction running on another context's stream (FramePrefetcher) a
    // tracker call stalled for the whole batch.  The arena only grows (between calls, to the previous call's high-water mark); a request
    // that does not fit falls back to hipMalloc for that call.
    char* arena = nullptr;
    size_t arena_cap = 0, arena_used = 0, arena_want = 0;
    std::vector<void*> arena_extra;
    // Pinned host staging of the per-frame fetch functions (pslfe_glue_fetch, pslfe_frame_fetch): a device-to-host copy into the caller's pageable
    // vectors blocks for ~20 us each, and a fetch is up to 13 of them; into pinned memory they are queued in ~2 us each and waited for once.
    char* hstage = nullptr;
    size_t hstage_cap = 0;

    int stage_begin(const char* name, hipEvent_t* a, hipEvent_t* b, hipStream_t on = nullptr);
    int stage_end(const char* name, hipEvent_t a, hipEvent_t b, hipStream_t on = nullptr);
    int resolve_pending();
};

// RAII-less helper used as: PSL_STAGE_BEGIN(ctx,"orb.fast"); launch...; PSL_STAGE_END(ctx,"orb.fast");
#define PSL_STAGE_BEGIN(ctx, name)                                  \
    hipEvent_t ev_a_ = nullptr, ev_b_ = nullptr;                    \
    const bool ev_on_ = (ctx)->profile && ((ctx)->profile_only.empty() || (ctx)->profile_only == (name)); \
    if (ev_on_) {                                                   \
        int rc_ = (ctx)->stage_begin(name, &ev_a_, &ev_b_);         \
        if (rc_) return rc_;                                        \
    }
#define PSL_STAGE_END(ctx, name)                                    \
    if (ev_on_) {                                                   \
        int rc_ = (ctx)->stage_end(name, ev_a_, ev_b_);             \
        if (rc_) return rc_;                                        \
    }

// same, for kernels launched on another stream of the context
#define PSL_STAGE_BEGIN_ON(ctx, name, st)                           \
    hipEvent_t ev_a_ = nullptr, ev_b_ = nullptr;                    \
    const bool ev_on_ = (ctx)->profile && ((ctx)->profile_only.empty() || (ctx)->profile_only == (name)); \
    if (ev_on_) {                                                   \
        int rc_ = (ctx)->stage_begin(name, &ev_a_, &ev_b_, st);     \
        if (rc_) return rc_;                                        \
    }
#define PSL_STAGE_END_ON(ctx, name, st)                             \
    if (ev_on_) {                                                   \
        int rc_ = (ctx)->stage_end(name, ev_a_, ev_b_, st);         \
        if (rc_) return rc_;                                        \
    }

static inline size_t psl_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// at least `bytes` of pinned host memory owned by the context (valid until the next call that asks for more); nullptr on failure
char* psl_host_stage(pslfe_ctx* ctx, size_t bytes);
// start of a call: releases the previous call's fall-back blocks and grows the arena to what that call wanted (calls on a context are serialised)
int psl_scratch_begin(pslfe_ctx* ctx);
// `bytes` of device memory valid until the next psl_scratch_begin on this context (256-byte aligned); nullptr on failure
void* psl_scratch(pslfe_ctx* ctx, size_t bytes);
template <typename T>
static inline T* psl_scratch_up(pslfe_ctx* ctx, const T* host, size_t count, hipStream_t st, hipError_t* e) {
    T* d = nullptr;
    if (*e == hipSuccess) {
        d = static_cast<T*>(psl_scratch(ctx, count ? count * sizeof(T) : 1));
        if (!d) *e = hipErrorOutOfMemory;
    }
    if (*e == hipSuccess && host && count) *e = hipMemcpyAsync(d, host, count * sizeof(T), hipMemcpyHostToDevice, st);
    return d;
}

#endif
