// libpslfe: Frame::ComputeBoW on the device. Product code.
// Reference behaviour reproduced: Frame::ComputeBoW src/Frame.cc:1053-1060 -> DBoW2 TemplatedVocabulary::transform(features,
// BowVector, FeatureVector, levelsup = 4) Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1124-1195 (TF_IDF weighting, L1_NORM
// scoring: the ORB vocabulary's settings) with the tree descent :1218-1260, BowVector::addWeight / normalize(L1)
// BowVector.cpp:34-84, FeatureVector::addFeature FeatureVector.cpp:31-45, FORB::distance FORB.cpp:81-101.
// The vocabulary is a host object (ORBvoc.txt); it is handed over once as flat arrays (pslfe_vocab_create).
//
// k_bow_descend: wave = descriptor; at every level the children of the current node are the lanes (Hamming distance, first
// minimum in child order = wave min of distance << 16 | child position).  k_bow_vectors: workgroup = frame; BowVector and
// FeatureVector are std::maps in the reference, i.e. ascending ids: features are ranked by (id, feature index) by counting,
// segment heads give the entries; a word's value is its weight added once per occurrence (f64, sequential), the L1 norm is
// the sequential f64 sum over ascending word ids, as the reference forms them.
#include <string.h>

#include <vector>

#include "pslfe_internal.h"

#define PSL_BOW_NMAX 4096

struct VocabDev {
    const int32_t* child_begin;
    const int32_t* child_count;
    const int32_t* child_ids;
    const uint32_t* node_desc;  // [nnodes][8]
    const double* node_weight;
    const int32_t* node_word;
    int nnodes, L;
};

__global__ __launch_bounds__(256) void k_bow_descend(VocabDev V, const uint8_t* __restrict__ desc, const int32_t* __restrict__ counts, int n_single,
                                                      int stride, int levelsup, int32_t* __restrict__ f_word, double* __restrict__ f_weight,
                                                      int32_t* __restrict__ f_nid) {
    const int frame = blockIdx.y, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = min(counts ? counts[frame] : n_single, stride);
    if (i >= n) return;
    const uint32_t* q = reinterpret_cast<const uint32_t*>(desc + ((size_t)frame * stride + i) * 32);
    uint32_t qd[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) qd[k] = q[k];
    const int nid_level = V.L - levelsup;
    int node = 0, nid = 0, level = 0;
    for (int guard = 0; guard < 64; ++guard) {
        const int cb = V.child_begin[node], cc = V.child_count[node];
        if (cc == 0) break;  // leaf
        ++level;
        uint32_t best = 0xffffffffu;
        for (int c = lane; c < cc; c += 64) {
            const int id = V.child_ids[cb + c];
            const uint4 d0 = *reinterpret_cast<const uint4*>(V.node_desc + (size_t)id * 8);
            const uint4 d1 = *reinterpret_cast<const uint4*>(V.node_desc + (size_t)id * 8 + 4);
            const int dist = __popc(qd[0] ^ d0.x) + __popc(qd[1] ^ d0.y) + __popc(qd[2] ^ d0.z) + __popc(qd[3] ^ d0.w) +
                             __popc(qd[4] ^ d1.x) + __popc(qd[5] ^ d1.y) + __popc(qd[6] ^ d1.z) + __popc(qd[7] ^ d1.w);
            const uint32_t key = ((uint32_t)dist << 16) | (uint32_t)min(c, 0xffff);
            best = key < best ? key : best;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint32_t u = __shfl_xor(best, o); best = u < best ? u : best; }
        node = V.child_ids[cb + (int)(best & 0xffff)];
        if (level == nid_level) nid = node;
    }
    if (lane == 0) {
        const size_t o = (size_t)frame * stride + i;
        f_word[o] = V.node_word[node];
        f_weight[o] = V.node_weight[node];
        f_nid[o] = nid;
    }
}

// exclusive block scan (256 threads)
__device__ __forceinline__ int psl_bow_scan(int v, int* s_w, int* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o); if (lane >= o) inc += u; }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    const int t0 = s_w[0], t1 = s_w[1], t2 = s_w[2], t3 = s_w[3];
    const int base = w == 0 ? 0 : (w == 1 ? t0 : (w == 2 ? t0 + t1 : t0 + t1 + t2));
    *total = t0 + t1 + t2 + t3;
    __syncthreads();
    return inc - v + base;
}

// One pass: rank the kept features by (key, feature index), find the segment heads, write the ascending ids, the start
// offsets and the member lists.  Returns the number of distinct ids (uniform).  s_key: LDS copy of the keys (kept: key >= 0).
__device__ int psl_bow_group(const int* s_key, int n, int* s_sorted /*LDS [n]: feature index by rank*/, int32_t* ids, int32_t* start, int32_t* members,
                             int* s_w, int* s_carry) {
    const int tid = threadIdx.x;
    int nkept = 0;
    for (int i = tid; i < n; i += 256) {
        const int k = s_key[i];
        if (k < 0) continue;
        int r = 0;
        for (int j = 0; j < n; ++j) { const int kj = s_key[j]; r += (kj >= 0) && (kj < k || (kj == k && j < i)); }
        s_sorted[r] = i;
    }
    for (int i = tid; i < n; i += 256) nkept += s_key[i] >= 0;
    // total kept (uniform)
    int tot;
    (void)psl_bow_scan(nkept, s_w, &tot);
    const int K = tot;
    if (tid == 0) *s_carry = 0;
    __syncthreads();
    int ngroups = 0;
    for (int base = 0; base < K; base += 256) {
        const int r = base + tid;
        bool head = false;
        int f = 0;
        if (r < K) {
            f = s_sorted[r];
            head = r == 0 || s_key[s_sorted[r - 1]] != s_key[f];
            if (members) members[r] = f;
        }
        int t;
        const int ex = psl_bow_scan(head ? 1 : 0, s_w, &t);
        const int g = *s_carry + ex;
        if (head) { ids[g] = s_key[f]; start[g] = r; }
        __syncthreads();
        if (tid == 0) *s_carry += t;
        __syncthreads();
    }
    ngroups = *s_carry;
    if (tid == 0) start[ngroups] = K;
    __syncthreads();
    return ngroups;
}

__global__ __launch_bounds__(256) void k_bow_vectors(const int32_t* __restrict__ counts, int n_single, int stride, const int32_t* __restrict__ f_word,
                                                      const double* __restrict__ f_weight, const int32_t* __restrict__ f_nid,
                                                      int32_t* __restrict__ bow_id, double* __restrict__ bow_val, int32_t* __restrict__ bow_start,
                                                      int32_t* __restrict__ nbow, int32_t* __restrict__ fv_node, int32_t* __restrict__ fv_start,
                                                      int32_t* __restrict__ fv_idx, int32_t* __restrict__ nfv) {
    __shared__ int s_key[PSL_BOW_NMAX];
    __shared__ int s_sorted[PSL_BOW_NMAX];
    __shared__ int s_w[4];
    __shared__ int s_carry;
    __shared__ double s_norm;
    const int frame = blockIdx.x, tid = threadIdx.x;
    const int n = min(min(counts ? counts[frame] : n_single, stride), PSL_BOW_NMAX);
    const size_t o = (size_t)frame * stride;
    // ---- BowVector (ascending word ids)
    for (int i = tid; i < n; i += 256) s_key[i] = f_weight[o + i] > 0 ? f_word[o + i] : -1;  // "if (w > 0) // not stopped"
    __syncthreads();
    int32_t* bstart = bow_start + (size_t)frame * (stride + 1);
    const int nb = psl_bow_group(s_key, n, s_sorted, bow_id + o, bstart, nullptr, s_w, &s_carry);
    for (int g = tid; g < nb; g += 256) {  // addWeight: the word's weight added once per occurrence
        const int f = s_sorted[bstart[g]];
        const double w = f_weight[o + f];
        double v = w;
        for (int t = bstart[g] + 1; t < bstart[g + 1]; ++t) v += w;
        bow_val[o + g] = v;
    }
    __syncthreads();
    if (tid == 0) {  // normalize(L1): sequential f64 sum over ascending ids
        double norm = 0.0;
        for (int g = 0; g < nb; ++g) norm += fabs(bow_val[o + g]);
        s_norm = norm;
    }
    __syncthreads();
    const double norm = s_norm;
    if (norm > 0.0)
        for (int g = tid; g < nb; g += 256) bow_val[o + g] = bow_val[o + g] / norm;
    if (tid == 0) nbow[frame] = nb;
    __syncthreads();
    // ---- FeatureVector (ascending node ids, feature indices ascending inside a node)
    for (int i = tid; i < n; i += 256) s_key[i] = f_weight[o + i] > 0 ? f_nid[o + i] : -1;
    __syncthreads();
    const int nf = psl_bow_group(s_key, n, s_sorted, fv_node + o, fv_start + (size_t)frame * (stride + 1), fv_idx + o, s_w, &s_carry);
    if (tid == 0) nfv[frame] = nf;
}

struct pslfe_vocab {
    pslfe_ctx* ctx = nullptr;
    VocabDev V = {};
    void* bufs[6] = {};
};

extern "C" {

void pslfe_vocab_destroy(pslfe_vocab* v) {
    if (!v) return;
    hipSetDevice(v->ctx->device);
    hipStreamSynchronize(v->ctx->stream);
    for (void* b : v->bufs) hipFree(b);
    delete v;
}

int pslfe_vocab_create(pslfe_ctx* ctx, int nnodes, const int32_t* child_begin, const int32_t* child_count, const int32_t* child_ids, int nchild,
                       const uint8_t* node_desc, const double* node_weight, const int32_t* node_word, int L, pslfe_vocab** out) {
    PSL_REQUIRE(ctx && out && child_begin && child_count && child_ids && node_desc && node_weight && node_word, PSLFE_E_INVALID,
                "pslfe_vocab_create: NULL argument");
    *out = nullptr;
    PSL_REQUIRE(nnodes >= 2 && nchild >= 1 && L >= 1 && L <= 32, PSLFE_E_INVALID, "pslfe_vocab_create: %d nodes, %d child links, L = %d", nnodes, nchild, L);
    PSL_REQUIRE(child_count[0] > 0, PSLFE_E_INVALID, "pslfe_vocab_create: the root has no children");
    for (int i = 0; i < nnodes; ++i) {
        PSL_REQUIRE(child_count[i] >= 0 && child_count[i] <= 65535 && child_begin[i] >= 0 && child_begin[i] + child_count[i] <= nchild, PSLFE_E_INVALID,
                    "pslfe_vocab_create: node %d has an invalid child range", i);
    }
    for (int i = 0; i < nchild; ++i)
        PSL_REQUIRE(child_ids[i] > 0 && child_ids[i] < nnodes, PSLFE_E_INVALID, "pslfe_vocab_create: child link %d -> node %d", i, child_ids[i]);
    {   // every leaf must be reached after at most L levels and the tree must not loop
        std::vector<int> depth(nnodes, -1);
        depth[0] = 0;
        std::vector<int> stack(1, 0);
        while (!stack.empty()) {
            const int nd = stack.back(); stack.pop_back();
            for (int c = 0; c < child_count[nd]; ++c) {
                const int ch = child_ids[child_begin[nd] + c];
                PSL_REQUIRE(depth[ch] < 0, PSLFE_E_INVALID, "pslfe_vocab_create: node %d has two parents", ch);
                depth[ch] = depth[nd] + 1;
                PSL_REQUIRE(depth[ch] <= L, PSLFE_E_INVALID, "pslfe_vocab_create: node %d lies deeper than L = %d", ch, L);
                stack.push_back(ch);
            }
        }
    }
    PSL_HIP(hipSetDevice(ctx->device));
    pslfe_vocab* v = new pslfe_vocab();
    v->ctx = ctx;
    const size_t sizes[6] = {(size_t)nnodes * 4, (size_t)nnodes * 4, (size_t)nchild * 4, (size_t)nnodes * 32, (size_t)nnodes * 8, (size_t)nnodes * 4};
    const void* srcs[6] = {child_begin, child_count, child_ids, node_desc, node_weight, node_word};
    hipError_t e = hipSuccess;
    for (int k = 0; k < 6 && e == hipSuccess; ++k) {
        e = hipMalloc(&v->bufs[k], sizes[k]);
        if (e == hipSuccess) e = hipMemcpy(v->bufs[k], srcs[k], sizes[k], hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        pslfe_set_error("pslfe_vocab_create: %s", hipGetErrorString(e));
        pslfe_vocab_destroy(v);
        return PSLFE_E_HIP;
    }
    v->V.child_begin = (const int32_t*)v->bufs[0]; v->V.child_count = (const int32_t*)v->bufs[1]; v->V.child_ids = (const int32_t*)v->bufs[2];
    v->V.node_desc = (const uint32_t*)v->bufs[3]; v->V.node_weight = (const double*)v->bufs[4]; v->V.node_word = (const int32_t*)v->bufs[5];
    v->V.nnodes = nnodes; v->V.L = L;
    *out = v;
    return PSLFE_OK;
}

int pslfe_compute_bow_device(pslfe_vocab* v, const uint8_t* d_desc, const int32_t* d_counts, int nframes, int stride, int levelsup,
                             int32_t* d_fword, double* d_fweight, int32_t* d_fnid, int32_t* d_bow_id, double* d_bow_val, int32_t* d_bow_start,
                             int32_t* d_nbow, int32_t* d_fv_node, int32_t* d_fv_start, int32_t* d_fv_idx, int32_t* d_nfv) {
    PSL_REQUIRE(v && d_desc && d_counts && d_fword && d_fweight && d_fnid && d_bow_id && d_bow_val && d_bow_start && d_nbow && d_fv_node &&
                    d_fv_start && d_fv_idx && d_nfv, PSLFE_E_INVALID, "pslfe_compute_bow_device: NULL argument");
    PSL_REQUIRE(nframes >= 1 && stride >= 1 && stride <= PSL_BOW_NMAX && levelsup >= 0, PSLFE_E_INVALID,
                "pslfe_compute_bow_device: %d frames, stride %d (max %d), levelsup %d", nframes, stride, PSL_BOW_NMAX, levelsup);
    PSL_HIP(hipSetDevice(v->ctx->device));
    hipStream_t st = v->ctx->stream;
    {
        PSL_STAGE_BEGIN(v->ctx, "bow.descend");
        k_bow_descend<<<dim3((stride + 3) / 4, nframes), 256, 0, st>>>(v->V, d_desc, d_counts, 0, stride, levelsup, d_fword, d_fweight, d_fnid);
        PSL_STAGE_END(v->ctx, "bow.descend");
    }
    {
        PSL_STAGE_BEGIN(v->ctx, "bow.vectors");
        k_bow_vectors<<<nframes, 256, 0, st>>>(d_counts, 0, stride, d_fword, d_fweight, d_fnid, d_bow_id, d_bow_val, d_bow_start, d_nbow, d_fv_node,
                                              d_fv_start, d_fv_idx, d_nfv);
        PSL_STAGE_END(v->ctx, "bow.vectors");
    }
    PSL_HIP(hipGetLastError());
    return PSLFE_OK;
}

int pslfe_compute_bow(pslfe_vocab* v, const uint8_t* desc, int n, int levelsup, int32_t* f_word, double* f_weight, int32_t* f_nid, int32_t* bow_id,
                      double* bow_val, int* nbow, int32_t* fv_node, int32_t* fv_start, int32_t* fv_idx, int* nfv) {
    PSL_REQUIRE(v && nbow && nfv && (n == 0 || desc), PSLFE_E_INVALID, "pslfe_compute_bow: NULL argument");
    PSL_REQUIRE(n >= 0 && n <= PSL_BOW_NMAX && levelsup >= 0, PSLFE_E_INVALID, "pslfe_compute_bow: %d features (max %d), levelsup %d", n, PSL_BOW_NMAX, levelsup);
    *nbow = 0; *nfv = 0;
    if (n == 0) { if (fv_start) fv_start[0] = 0; return PSLFE_OK; }
    PSL_HIP(hipSetDevice(v->ctx->device));
    hipStream_t st = v->ctx->stream;
    // one allocation: desc | fword | fnid | bow_id | bow_start | fv_node | fv_start | fv_idx | counters | fweight | bow_val
    const size_t N = (size_t)n, i4 = 4, need = N * 32 + (N * 5 + 2 * (N + 1) + 4) * i4 + 16 + N * 16;
    { const int rc_ = psl_scratch_begin(v->ctx); if (rc_) return rc_; }
    uint8_t* base = static_cast<uint8_t*>(psl_scratch(v->ctx, need));   // the context's scratch arena (no hipMalloc / hipFree per call)
    PSL_REQUIRE(base, PSLFE_E_HIP, "pslfe_compute_bow: out of device memory");
    uint8_t* p = base;
    uint8_t* d_desc = p; p += N * 32;
    int32_t* d_fword = (int32_t*)p; p += N * 4;
    int32_t* d_fnid = (int32_t*)p; p += N * 4;
    int32_t* d_bow_id = (int32_t*)p; p += N * 4;
    int32_t* d_bow_start = (int32_t*)p; p += (N + 1) * 4;
    int32_t* d_fv_node = (int32_t*)p; p += N * 4;
    int32_t* d_fv_start = (int32_t*)p; p += (N + 1) * 4;
    int32_t* d_fv_idx = (int32_t*)p; p += N * 4;
    int32_t* d_cnt = (int32_t*)p; p += 4 * 4;  // [0] = nbow, [1] = nfv
    p = (uint8_t*)(((uintptr_t)p + 15) & ~(uintptr_t)15);
    double* d_fweight = (double*)p; p += N * 8;
    double* d_bow_val = (double*)p;
    int rc = PSLFE_OK;
    hipError_t e = hipMemcpyAsync(d_desc, desc, N * 32, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        k_bow_descend<<<dim3((n + 3) / 4, 1), 256, 0, st>>>(v->V, d_desc, nullptr, n, n, levelsup, d_fword, d_fweight, d_fnid);
        k_bow_vectors<<<1, 256, 0, st>>>(nullptr, n, n, d_fword, d_fweight, d_fnid, d_bow_id, d_bow_val, d_bow_start, d_cnt, d_fv_node, d_fv_start,
                                        d_fv_idx, d_cnt + 1);
        e = hipGetLastError();
    }
    int cnt[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(cnt, d_cnt, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    auto D = [&](void* dst, const void* src, size_t bytes) { if (e == hipSuccess && dst && bytes) e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st); };
    D(f_word, d_fword, N * 4); D(f_weight, d_fweight, N * 8); D(f_nid, d_fnid, N * 4);
    D(bow_id, d_bow_id, (size_t)cnt[0] * 4); D(bow_val, d_bow_val, (size_t)cnt[0] * 8);
    D(fv_node, d_fv_node, (size_t)cnt[1] * 4); D(fv_start, d_fv_start, ((size_t)cnt[1] + 1) * 4);
    if (e == hipSuccess && fv_idx && cnt[1] > 0) {
        int kept = 0;
        e = hipMemcpyAsync(&kept, d_fv_start + cnt[1], 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        D(fv_idx, d_fv_idx, (size_t)kept * 4);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { pslfe_set_error("pslfe_compute_bow: %s", hipGetErrorString(e)); rc = PSLFE_E_HIP; }
    hipStreamSynchronize(st);
    *nbow = cnt[0]; *nfv = cnt[1];
    return rc;
}

}  // extern "C"
