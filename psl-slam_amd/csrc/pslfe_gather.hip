// libpslfe: the batched many-frames mode across the GPUs of one node (BASELINE config 4). Product code.
// Frames and whole streams are independent - stream s runs on rank s mod world, no data-path collective.  The one exchange
// is the RESULT GATHER: every rank packs the results of its batch into fixed-size per-frame records and the records are
// gathered with RCCL over xGMI - to the consuming rank (pslfe_gather_to_root: one group of ncclSend / ncclRecv, only the root
// holds world x batch records) or to every rank (pslfe_gather_all: ncclAllGather) -, one communicator per context, on the
// gather's own stream so that the next batch's kernels overlap the exchange.  Record = everything Tracking.cc reads of a Frame on this path:
//   {counts, mvKeys, mDescriptors, point matches, mvKeylinesUn, mLdesc, mvKeyLineFunctions, line matches, fans, mvPlanes,
//    mvPlaneLineNo}; ~85 KB at 1000 points / 200 lines, i.e. 3.4 GB/s per rank at 40 k frames/s against 153 GB/s per xGMI link:
//   the exchange is latency, not bandwidth - hence ONE collective per batch over all records, not one per frame or per array.
// RCCL is loaded at run time (dlopen of librccl.so.1: the copy a host process already holds, e.g. PyTorch's, or ROCm's), so the
// library itself has no link-time dependency on it and loads on machines without RCCL.
#include <dlfcn.h>
#include <string.h>

#include "pslfe_internal.h"

namespace {

inline int64_t align16(int64_t v) { return (v + 15) / 16 * 16; }

struct Section { int64_t off, row_bytes; };

struct PackArgs {
    PslRecordCaps caps;
    PslRecordLayout L;
    PslRecordSources S;
};

// workgroup = frame: header + sections, 4-byte words (every row size is a multiple of 4)
__device__ __forceinline__ void copy_words(uint8_t* dst, const uint8_t* src, int64_t bytes, int tid) {
    const uint32_t* s = reinterpret_cast<const uint32_t*>(src);
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
    for (int64_t i = tid; i < bytes / 4; i += 256) d[i] = s[i];
}

// match / lmatch: `cap` int32 rows indexed by the query; rows the caller's buffer does not hold (stride < cap) read "no match" = -1
__device__ __forceinline__ void copy_match(uint8_t* dst, const int32_t* src, int cap, int stride, int tid) {
    int32_t* d = reinterpret_cast<int32_t*>(dst);
    const int have = min(cap, stride);
    for (int i = tid; i < cap; i += 256) d[i] = i < have ? src[i] : -1;
}

__global__ __launch_bounds__(256) void k_record_pack(PackArgs A, uint8_t* __restrict__ out) {
    const int f = blockIdx.x, tid = threadIdx.x;
    uint8_t* rec = out + (size_t)f * A.L.bytes;
    const PslRecordSources& S = A.S;
    const int n_kp = S.d_kp_counts ? S.d_kp_counts[f] : 0;
    const int n_match = S.d_nmatches ? S.d_nmatches[f] : 0;
    const int n_kl = S.d_kl_counts ? S.d_kl_counts[f] : 0;
    const int n_lmatch = S.d_nlmatches ? S.d_nlmatches[f] : 0;
    const int n_fan = S.d_fan_counts ? S.d_fan_counts[f] : 0;
    const int n_pl = S.d_plane_counts ? S.d_plane_counts[f] : 0;
    const int c_kp = min(n_kp, A.caps.kp_cap), c_kl = min(n_kl, A.caps.kl_cap), c_fan = min(n_fan, A.caps.fan_cap), c_pl = min(n_pl, A.caps.plane_cap);
    if (tid == 0) {
        int32_t* h = reinterpret_cast<int32_t*>(rec);
        h[0] = n_kp; h[1] = n_match; h[2] = n_kl; h[3] = n_lmatch; h[4] = n_fan; h[5] = n_pl;
        h[6] = (n_kp > c_kp ? 1 : 0) | (n_kl > c_kl ? 2 : 0) | (n_fan > c_fan ? 4 : 0) | (n_pl > c_pl ? 8 : 0);
        h[7] = f;
    }
    if (S.d_kps) copy_words(rec + A.L.off_kps, reinterpret_cast<const uint8_t*>(S.d_kps) + (size_t)f * S.kp_stride * 28, (int64_t)c_kp * 28, tid);
    if (S.d_desc) copy_words(rec + A.L.off_desc, S.d_desc + (size_t)f * S.kp_stride * 32, (int64_t)c_kp * 32, tid);
    if (S.d_match) copy_match(rec + A.L.off_match, S.d_match + (size_t)f * S.match_stride, A.caps.kp_cap, S.match_stride, tid);  // indexed by the query (= a keypoint of the previous frame): fixed size
    if (S.d_kls) copy_words(rec + A.L.off_kls, reinterpret_cast<const uint8_t*>(S.d_kls) + (size_t)f * S.kl_stride * 68, (int64_t)c_kl * 68, tid);
    if (S.d_ldesc) copy_words(rec + A.L.off_ldesc, S.d_ldesc + (size_t)f * S.kl_stride * 32, (int64_t)c_kl * 32, tid);
    if (S.d_lineEq) copy_words(rec + A.L.off_lineEq, reinterpret_cast<const uint8_t*>(S.d_lineEq) + (size_t)f * S.kl_stride * 24, (int64_t)c_kl * 24, tid);
    if (S.d_lmatch) copy_match(rec + A.L.off_lmatch, S.d_lmatch + (size_t)f * S.lmatch_stride, A.caps.kl_cap, S.lmatch_stride, tid);
    if (S.d_fans) copy_words(rec + A.L.off_fans, reinterpret_cast<const uint8_t*>(S.d_fans) + (size_t)f * S.fan_stride * 16, (int64_t)c_fan * 16, tid);
    if (S.d_planes) copy_words(rec + A.L.off_planes, reinterpret_cast<const uint8_t*>(S.d_planes) + (size_t)f * S.plane_stride * 16, (int64_t)c_pl * 16, tid);
    if (S.d_plane_lines) copy_words(rec + A.L.off_plane_lines, reinterpret_cast<const uint8_t*>(S.d_plane_lines) + (size_t)f * S.plane_stride * 8, (int64_t)c_pl * 8, tid);
}

// ---- RCCL through dlopen -----------------------------------------------------------------------------------------------
struct NcclId { char internal[128]; };
typedef int (*fn_get_id)(NcclId*);
typedef int (*fn_init_rank)(void**, int, NcclId, int);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_send)(const void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_recv)(void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_group)(void);
typedef int (*fn_destroy)(void*);
typedef const char* (*fn_errstr)(int);

struct Rccl {
    void* h = nullptr;
    fn_get_id get_id = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_send send = nullptr;            // ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd: pslfe_gather_to_root
    fn_recv recv = nullptr;
    fn_group group_start = nullptr, group_end = nullptr;
    fn_destroy destroy = nullptr;
    fn_errstr errstr = nullptr;
};

Rccl* rccl() {
    static Rccl R;
    static bool tried = false;
    if (!tried) {
        tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            R.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (R.h) break;
        }
        if (R.h) {
            R.get_id = (fn_get_id)dlsym(R.h, "ncclGetUniqueId");
            R.init_rank = (fn_init_rank)dlsym(R.h, "ncclCommInitRank");
            R.all_gather = (fn_all_gather)dlsym(R.h, "ncclAllGather");
            R.send = (fn_send)dlsym(R.h, "ncclSend");
            R.recv = (fn_recv)dlsym(R.h, "ncclRecv");
            R.group_start = (fn_group)dlsym(R.h, "ncclGroupStart");
            R.group_end = (fn_group)dlsym(R.h, "ncclGroupEnd");
            R.destroy = (fn_destroy)dlsym(R.h, "ncclCommDestroy");
            R.errstr = (fn_errstr)dlsym(R.h, "ncclGetErrorString");
            if (!R.get_id || !R.init_rank || !R.all_gather || !R.destroy) { dlclose(R.h); R.h = nullptr; }
        }
    }
    return R.h ? &R : nullptr;
}

#define PSL_NCCL(R, call)                                                                                     \
    do {                                                                                                      \
        const int e_ = (call);                                                                                \
        if (e_ != 0) {                                                                                        \
            pslfe_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, (R)->errstr ? (R)->errstr(e_) : "RCCL error"); \
            return PSLFE_E_HIP;                                                                               \
        }                                                                                                     \
    } while (0)

}  // namespace

struct pslfe_gather {
    pslfe_ctx* ctx = nullptr;
    int rank = 0, world = 1;
    void* comm = nullptr;
    hipStream_t stream = nullptr;   // the exchange runs here, ordered after the context's stream by ev_ready
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    bool pending = false;        // an exchange has been issued whose completion the HOST has not waited for
    bool stream_waited = false;  // ... but the context's stream already waits for it (pslfe_gather_wait(g, 0))
};

extern "C" {

int pslfe_record_layout(const PslRecordCaps* caps, PslRecordLayout* out) {
    PSL_REQUIRE(caps && out, PSLFE_E_INVALID, "pslfe_record_layout: NULL argument");
    PSL_REQUIRE(caps->kp_cap >= 0 && caps->kl_cap >= 0 && caps->fan_cap >= 0 && caps->plane_cap >= 0 && caps->kp_cap <= (1 << 20) &&
                caps->kl_cap <= (1 << 20) && caps->fan_cap <= (1 << 20) && caps->plane_cap <= (1 << 20), PSLFE_E_INVALID, "pslfe_record_layout: capacities");
    int64_t o = 32;  // header: 8 x int32
    out->off_kps = o;         o = align16(o + (int64_t)caps->kp_cap * 28);
    out->off_desc = o;        o = align16(o + (int64_t)caps->kp_cap * 32);
    out->off_match = o;       o = align16(o + (int64_t)caps->kp_cap * 4);
    out->off_kls = o;         o = align16(o + (int64_t)caps->kl_cap * 68);
    out->off_ldesc = o;       o = align16(o + (int64_t)caps->kl_cap * 32);
    out->off_lineEq = o;      o = align16(o + (int64_t)caps->kl_cap * 24);
    out->off_lmatch = o;      o = align16(o + (int64_t)caps->kl_cap * 4);
    out->off_fans = o;        o = align16(o + (int64_t)caps->fan_cap * 16);
    out->off_planes = o;      o = align16(o + (int64_t)caps->plane_cap * 16);
    out->off_plane_lines = o; o = align16(o + (int64_t)caps->plane_cap * 8);
    out->bytes = (o + 255) / 256 * 256;
    return PSLFE_OK;
}

int pslfe_record_pack_device(pslfe_ctx* ctx, const PslRecordCaps* caps, const PslRecordSources* src, int nframes, void* d_records) {
    PSL_REQUIRE(ctx && caps && src && d_records && nframes >= 1, PSLFE_E_INVALID, "pslfe_record_pack_device: bad argument");
    PackArgs A;
    A.caps = *caps; A.S = *src;
    int rc = pslfe_record_layout(caps, &A.L);
    if (rc) return rc;
    PSL_REQUIRE((!src->d_kps && !src->d_desc) || src->kp_stride >= caps->kp_cap, PSLFE_E_INVALID, "pslfe_record_pack_device: kp_stride %d < kp_cap %d", src->kp_stride, caps->kp_cap);
    PSL_REQUIRE((!src->d_kls && !src->d_ldesc && !src->d_lineEq) || src->kl_stride >= caps->kl_cap, PSLFE_E_INVALID, "pslfe_record_pack_device: kl_stride %d < kl_cap %d", src->kl_stride, caps->kl_cap);
    PSL_REQUIRE(!src->d_fans || src->fan_stride >= caps->fan_cap, PSLFE_E_INVALID, "pslfe_record_pack_device: fan_stride %d < fan_cap %d", src->fan_stride, caps->fan_cap);
    PSL_REQUIRE((!src->d_planes && !src->d_plane_lines) || src->plane_stride >= caps->plane_cap, PSLFE_E_INVALID, "pslfe_record_pack_device: plane_stride %d < plane_cap %d", src->plane_stride, caps->plane_cap);
    PSL_HIP(hipSetDevice(ctx->device));
    PSL_HIP(hipMemsetAsync(d_records, 0, (size_t)nframes * A.L.bytes, ctx->stream));  // padding is defined: records compare bytewise
    {
        PSL_STAGE_BEGIN(ctx, "gather.pack");
        k_record_pack<<<nframes, 256, 0, ctx->stream>>>(A, static_cast<uint8_t*>(d_records));
        PSL_STAGE_END(ctx, "gather.pack");
    }
    PSL_HIP(hipGetLastError());
    return PSLFE_OK;
}

int pslfe_gather_unique_id(uint8_t id[128]) {
    PSL_REQUIRE(id, PSLFE_E_INVALID, "pslfe_gather_unique_id: NULL argument");
    Rccl* R = rccl();
    PSL_REQUIRE(R, PSLFE_E_NODEVICE, "pslfe_gather: librccl.so.1 could not be loaded");
    NcclId u;
    PSL_NCCL(R, R->get_id(&u));
    memcpy(id, u.internal, 128);
    return PSLFE_OK;
}

int pslfe_gather_create(pslfe_ctx* ctx, int rank, int world, const uint8_t id[128], pslfe_gather** out) {
    PSL_REQUIRE(ctx && id && out && world >= 1 && rank >= 0 && rank < world, PSLFE_E_INVALID, "pslfe_gather_create: rank %d of %d", rank, world);
    *out = nullptr;
    Rccl* R = rccl();
    PSL_REQUIRE(R, PSLFE_E_NODEVICE, "pslfe_gather: librccl.so.1 could not be loaded");
    PSL_HIP(hipSetDevice(ctx->device));
    pslfe_gather* g = new pslfe_gather();
    g->ctx = ctx; g->rank = rank; g->world = world;
    NcclId u;
    memcpy(u.internal, id, 128);
    const int e = R->init_rank(&g->comm, world, u, rank);
    if (e != 0) {
        pslfe_set_error("pslfe_gather_create: ncclCommInitRank -> %s", R->errstr ? R->errstr(e) : "RCCL error");
        delete g;
        return PSLFE_E_HIP;
    }
    if (hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&g->ev_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g->ev_done, hipEventDisableTiming) != hipSuccess) {
        pslfe_set_error("pslfe_gather_create: stream / event creation failed");
        pslfe_gather_destroy(g);
        return PSLFE_E_HIP;
    }
    *out = g;
    return PSLFE_OK;
}

void pslfe_gather_destroy(pslfe_gather* g) {
    if (!g) return;
    hipSetDevice(g->ctx->device);
    if (g->stream) hipStreamSynchronize(g->stream);
    Rccl* R = rccl();
    if (R && g->comm) R->destroy(g->comm);
    if (g->ev_ready) hipEventDestroy(g->ev_ready);
    if (g->ev_done) hipEventDestroy(g->ev_done);
    if (g->stream) hipStreamDestroy(g->stream);
    delete g;
}

int pslfe_gather_all(pslfe_gather* g, const void* d_send, size_t bytes_per_rank, void* d_recv) {
    PSL_REQUIRE(g && d_send && d_recv && bytes_per_rank > 0, PSLFE_E_INVALID, "pslfe_gather_all: bad argument");
    Rccl* R = rccl();
    PSL_REQUIRE(R, PSLFE_E_NODEVICE, "pslfe_gather: librccl.so.1 could not be loaded");
    PSL_HIP(hipSetDevice(g->ctx->device));
    // ordered after everything issued on the context's stream so far (the pack kernel), but on its own stream: the next batch's
    // kernels do not wait for the exchange
    PSL_HIP(hipEventRecord(g->ev_ready, g->ctx->stream));
    PSL_HIP(hipStreamWaitEvent(g->stream, g->ev_ready, 0));
    PSL_NCCL(R, R->all_gather(d_send, d_recv, bytes_per_rank, /* ncclInt8 */ 0, g->comm, g->stream));
    PSL_HIP(hipEventRecord(g->ev_done, g->stream));
    g->pending = true;
    g->stream_waited = false;
    return PSLFE_OK;
}

int pslfe_gather_to_root(pslfe_gather* g, const void* d_send, size_t bytes_per_rank, int root, void* d_recv) {
    PSL_REQUIRE(g && d_send && bytes_per_rank > 0 && root >= 0 && root < g->world, PSLFE_E_INVALID, "pslfe_gather_to_root: bad argument (root %d of %d)", root, g ? g->world : 0);
    PSL_REQUIRE(g->rank != root || d_recv, PSLFE_E_INVALID, "pslfe_gather_to_root: the root needs a receive buffer");
    Rccl* R = rccl();
    PSL_REQUIRE(R, PSLFE_E_NODEVICE, "pslfe_gather: librccl.so.1 could not be loaded");
    PSL_REQUIRE(R->send && R->recv && R->group_start && R->group_end, PSLFE_E_NODEVICE, "pslfe_gather_to_root: this RCCL has no ncclSend / ncclRecv");
    PSL_HIP(hipSetDevice(g->ctx->device));
    PSL_HIP(hipEventRecord(g->ev_ready, g->ctx->stream));
    PSL_HIP(hipStreamWaitEvent(g->stream, g->ev_ready, 0));
    // ONE group: every rank sends its records to the root (the root to itself: RCCL turns a self send / recv pair into a device
    // copy), the root posts one receive per rank.  A failure inside the group still closes it, so the communicator stays usable.
    PSL_NCCL(R, R->group_start());
    int e = R->send(d_send, bytes_per_rank, /* ncclInt8 */ 0, root, g->comm, g->stream);
    if (e == 0 && g->rank == root)
        for (int r = 0; r < g->world && e == 0; ++r)
            e = R->recv(static_cast<uint8_t*>(d_recv) + (size_t)r * bytes_per_rank, bytes_per_rank, /* ncclInt8 */ 0, r, g->comm, g->stream);
    const int e2 = R->group_end();
    if (e != 0 || e2 != 0) {
        pslfe_set_error("pslfe_gather_to_root: ncclSend / ncclRecv -> %s", R->errstr ? R->errstr(e != 0 ? e : e2) : "RCCL error");
        return PSLFE_E_HIP;
    }
    PSL_HIP(hipEventRecord(g->ev_done, g->stream));
    g->pending = true;
    g->stream_waited = false;
    return PSLFE_OK;
}

int pslfe_gather_wait(pslfe_gather* g, int host_blocking) {
    PSL_REQUIRE(g, PSLFE_E_INVALID, "pslfe_gather_wait: gather is NULL");
    if (!g->pending) return PSLFE_OK;
    PSL_HIP(hipSetDevice(g->ctx->device));
    if (host_blocking) {   // the host waits; only this clears `pending` (a stream-side wait before it does not make the host safe)
        PSL_HIP(hipEventSynchronize(g->ev_done));
        g->pending = false;
    } else if (!g->stream_waited) {
        PSL_HIP(hipStreamWaitEvent(g->ctx->stream, g->ev_done, 0));
        g->stream_waited = true;
    }
    return PSLFE_OK;
}

int pslfe_gather_world(const pslfe_gather* g, int* rank, int* world) {
    PSL_REQUIRE(g, PSLFE_E_INVALID, "pslfe_gather_world: gather is NULL");
    if (rank) *rank = g->rank;
    if (world) *world = g->world;
    return PSLFE_OK;
}

}  // extern "C"
