// libpslfe: context, error reporting, stage timing. Product code.
#include <stdarg.h>
#include <string.h>

#include "pslfe_internal.h"

static thread_local char g_err[512] = "";

void pslfe_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int pslfe_ctx::stage_begin(const char* name, hipEvent_t* a, hipEvent_t* b, hipStream_t on) {
    (void)name;
    PSL_HIP(hipEventCreate(a));
    PSL_HIP(hipEventCreate(b));
    PSL_HIP(hipEventRecord(*a, on ? on : stream));
    return PSLFE_OK;
}

int pslfe_ctx::stage_end(const char* name, hipEvent_t a, hipEvent_t b, hipStream_t on) {
    PSL_HIP(hipEventRecord(b, on ? on : stream));
    pending.push_back({a, b, name});
    return PSLFE_OK;
}

int pslfe_ctx::resolve_pending() {
    for (auto& p : pending) {
        PSL_HIP(hipEventSynchronize(p.b));
        float ms = 0;
        PSL_HIP(hipEventElapsedTime(&ms, p.a, p.b));
        StageTimer& t = stages[p.stage];
        t.ms += ms;
        t.launches += 1;
        hipEventDestroy(p.a);
        hipEventDestroy(p.b);
    }
    pending.clear();
    return PSLFE_OK;
}

char* psl_host_stage(pslfe_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->hstage_cap) return ctx->hstage;
    if (ctx->hstage) { hipHostFree(ctx->hstage); ctx->hstage = nullptr; ctx->hstage_cap = 0; }
    const size_t want = psl_align_up(bytes + bytes / 2, (size_t)1 << 16);
    void* q = nullptr;
    if (hipHostMalloc(&q, want, hipHostMallocDefault) != hipSuccess) return nullptr;
    ctx->hstage = static_cast<char*>(q); ctx->hstage_cap = want;
    return ctx->hstage;
}

int psl_scratch_begin(pslfe_ctx* ctx) {
    if (!ctx->arena_extra.empty()) {   // the previous call outgrew the arena: its fall-back blocks go, and the arena grows
        PSL_HIP(hipStreamSynchronize(ctx->stream));
        for (void* q : ctx->arena_extra) hipFree(q);
        ctx->arena_extra.clear();
    }
    if (ctx->arena_want > ctx->arena_cap) {
        PSL_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->arena) hipFree(ctx->arena);
        ctx->arena = nullptr; ctx->arena_cap = 0;
        const size_t want = psl_align_up(ctx->arena_want + ctx->arena_want / 2, 1 << 20);
        PSL_HIP(hipMalloc((void**)&ctx->arena, want));
        ctx->arena_cap = want;
    }
    ctx->arena_used = 0; ctx->arena_want = 0;
    return PSLFE_OK;
}

void* psl_scratch(pslfe_ctx* ctx, size_t bytes) {
    const size_t b = psl_align_up(bytes ? bytes : 1, 256);
    ctx->arena_want += b;
    if (ctx->arena && ctx->arena_used + b <= ctx->arena_cap) {
        void* p = ctx->arena + ctx->arena_used;
        ctx->arena_used += b;
        return p;
    }
    void* p = nullptr;
    if (hipMalloc(&p, b) != hipSuccess) return nullptr;
    ctx->arena_extra.push_back(p);
    return p;
}

extern "C" {

const char* pslfe_version(void) { return "pslfe 0.1 (gfx950)"; }
const char* pslfe_last_error(void) { return g_err; }

int pslfe_ctx_create(int device, pslfe_ctx** out) {
    PSL_REQUIRE(out, PSLFE_E_INVALID, "pslfe_ctx_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        pslfe_set_error("pslfe_ctx_create: no HIP device (%s); this library has no CPU fallback",
                        e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return PSLFE_E_NODEVICE;
    }
    PSL_REQUIRE(device >= 0 && device < ndev, PSLFE_E_INVALID, "pslfe_ctx_create: device %d of %d", device, ndev);
    if (hipSetDevice(device) != hipSuccess) {
        pslfe_set_error("pslfe_ctx_create: hipSetDevice(%d) failed", device);
        return PSLFE_E_NODEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        pslfe_set_error("pslfe_ctx_create: hipGetDeviceProperties failed");
        return PSLFE_E_NODEVICE;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        pslfe_set_error("pslfe_ctx_create: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return PSLFE_E_NODEVICE;
    }
    pslfe_ctx* c = new pslfe_ctx();
    c->device = device;
    c->cu_count = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        pslfe_set_error("pslfe_ctx_create: hipStreamCreate failed");
        return PSLFE_E_HIP;
    }
    c->stream = c->own_stream;
    if (hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
        pslfe_ctx_destroy(c);
        pslfe_set_error("pslfe_ctx_create: auxiliary stream / events failed");
        return PSLFE_E_HIP;
    }
    *out = c;
    return PSLFE_OK;
}

void pslfe_ctx_destroy(pslfe_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (void* q : ctx->arena_extra) hipFree(q);
    if (ctx->arena) hipFree(ctx->arena);
    if (ctx->hstage) hipHostFree(ctx->hstage);
    if (ctx->aux_stream) hipStreamSynchronize(ctx->aux_stream);
    ctx->resolve_pending();
    if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
    if (ctx->aux_stream) hipStreamDestroy(ctx->aux_stream);
    if (ctx->own_stream) hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int pslfe_ctx_set_stream(pslfe_ctx* ctx, void* hip_stream) {
    PSL_REQUIRE(ctx, PSLFE_E_INVALID, "pslfe_ctx_set_stream: ctx is NULL");
    PSL_HIP(hipSetDevice(ctx->device));
    PSL_HIP(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return PSLFE_OK;
}

int pslfe_ctx_synchronize(pslfe_ctx* ctx) {
    PSL_REQUIRE(ctx, PSLFE_E_INVALID, "pslfe_ctx_synchronize: ctx is NULL");
    PSL_HIP(hipSetDevice(ctx->device));
    PSL_HIP(hipStreamSynchronize(ctx->stream));
    return ctx->resolve_pending();
}

int pslfe_ctx_profile(pslfe_ctx* ctx, int enable) {
    PSL_REQUIRE(ctx, PSLFE_E_INVALID, "pslfe_ctx_profile: ctx is NULL");
    int rc = pslfe_ctx_synchronize(ctx);
    if (rc) return rc;
    ctx->profile = enable != 0;
    return PSLFE_OK;
}

int pslfe_ctx_profile_only(pslfe_ctx* ctx, const char* stage) {
    PSL_REQUIRE(ctx, PSLFE_E_INVALID, "pslfe_ctx_profile_only: ctx is NULL");
    int rc = pslfe_ctx_synchronize(ctx);
    if (rc) return rc;
    ctx->profile_only = stage ? stage : "";
    return PSLFE_OK;
}

int pslfe_ctx_profile_reset(pslfe_ctx* ctx) {
    PSL_REQUIRE(ctx, PSLFE_E_INVALID, "pslfe_ctx_profile_reset: ctx is NULL");
    int rc = pslfe_ctx_synchronize(ctx);
    if (rc) return rc;
    ctx->stages.clear();
    return PSLFE_OK;
}

int pslfe_ctx_stage_time(pslfe_ctx* ctx, const char* stage, double* ms_total, int* launches) {
    PSL_REQUIRE(ctx && stage, PSLFE_E_INVALID, "pslfe_ctx_stage_time: NULL argument");
    int rc = pslfe_ctx_synchronize(ctx);
    if (rc) return rc;
    auto it = ctx->stages.find(stage);
    if (ms_total) *ms_total = it == ctx->stages.end() ? 0.0 : it->second.ms;
    if (launches) *launches = it == ctx->stages.end() ? 0 : it->second.launches;
    return PSLFE_OK;
}


// ---- plain HBM helpers for callers that do not bring their own device allocator ------------------
int pslfe_device_alloc(pslfe_ctx* ctx, size_t bytes, void** d_ptr) {
    PSL_REQUIRE(ctx && d_ptr, PSLFE_E_INVALID, "pslfe_device_alloc: NULL argument");
    PSL_HIP(hipSetDevice(ctx->device));
    PSL_HIP(hipMalloc(d_ptr, bytes ? bytes : 1));
    return PSLFE_OK;
}

int pslfe_device_free(pslfe_ctx* ctx, void* d_ptr) {
    PSL_REQUIRE(ctx, PSLFE_E_INVALID, "pslfe_device_free: ctx is NULL");
    PSL_HIP(hipSetDevice(ctx->device));
    PSL_HIP(hipStreamSynchronize(ctx->stream));
    PSL_HIP(hipFree(d_ptr));
    return PSLFE_OK;
}

int pslfe_device_upload(pslfe_ctx* ctx, void* d_dst, const void* src, size_t bytes) {
    PSL_REQUIRE(ctx && d_dst && src, PSLFE_E_INVALID, "pslfe_device_upload: NULL argument");
    PSL_HIP(hipSetDevice(ctx->device));
    PSL_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    PSL_HIP(hipStreamSynchronize(ctx->stream));
    return PSLFE_OK;
}

int pslfe_device_download(pslfe_ctx* ctx, void* dst, const void* d_src, size_t bytes) {
    PSL_REQUIRE(ctx && dst && d_src, PSLFE_E_INVALID, "pslfe_device_download: NULL argument");
    PSL_HIP(hipSetDevice(ctx->device));
    PSL_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PSL_HIP(hipStreamSynchronize(ctx->stream));
    return PSLFE_OK;
}

}  // extern "C"
