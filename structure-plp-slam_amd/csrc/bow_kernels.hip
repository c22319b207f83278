// Bag-of-words transform of a frame's ORB descriptors (data::frame::compute_bow, src/PLPSLAM/data/frame.cc:785-795:
// bow_vocab_->transform(descriptors, bow_vec_, bow_feat_vec_, 4) with DBoW2's TemplatedVocabulary) for B frames at once.
//
//   k_bow_descend   16 lanes per descriptor (4 per wave): lane c compares the descriptor with child c of the current node
//                   (the children of one node are read as one 16 x 32 B gather), the row's minimum of (distance, child
//                   position) is the first-minimum child of the reference's loop; the node passed at level L - levelsup
//                   is the feature's NodeId.  grid = (ceil(cap / 16), B), block = 256.
//   k_bow_assemble  one workgroup per frame turns the per-feature (word, weight, node) triples into the two ordered maps:
//                   bitonic sort of (word << 32 | feature) in LDS, run heads = distinct words, the weight added once per
//                   feature in feature order (a run of r features is r - 1 sequential additions), the L1 / L2 norm as ONE
//                   sequential f64 chain in word order (that is what std::map iteration does), then the same sort on
//                   (node << 32 | feature) for the FeatureVector.  grid = B, block = 256, LDS = 16 B x pow2(cap).
// Everything is integer work plus IEEE f64 +, *, /, sqrt -> bit-exact against the oracle.
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "plp_common.hpp"
#include "plp_barrier.hpp"

namespace plp {

struct BowTreeDev {
    const int32_t* child_offset; const int32_t* children; const uint8_t* node_desc; const double* node_weight; const uint32_t* node_word;
    int L;
};

namespace {

__device__ __forceinline__ uint32_t row16_min_u32(uint32_t v) {
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xf, 0xf, false));   // row_ror:8
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124, 0xf, 0xf, false));   // row_ror:4
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x122, 0xf, 0xf, false));   // row_ror:2
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x121, 0xf, 0xf, false));   // row_ror:1
    return v;
}

__global__ __launch_bounds__(256) void k_bow_descend(BowTreeDev V, const uint8_t* __restrict__ desc, const int32_t* __restrict__ counts, int cap,
                                                     int levelsup, uint32_t* __restrict__ word_id, uint32_t* __restrict__ node_id,
                                                     double* __restrict__ weight) {
    const int b = blockIdx.y, f = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    const int n = counts ? min(counts[b], cap) : cap;
    if (f >= n) return;                                   // whole 16-lane rows leave together
    const size_t o = (size_t)b * cap + f;
    const uint4* fp = reinterpret_cast<const uint4*>(desc + o * 32);
    const uint4 f0 = fp[0], f1 = fp[1];
    const int nid_level = V.L - levelsup;
    uint32_t node = 0, nid = 0;
    bool nid_set = nid_level <= 0;
    int level = 0;
    int off = V.child_offset[0], nc = V.child_offset[1] - off;
    while (nc > 0) {
        ++level;
        uint32_t best = 0xFFFFFFFFu;
        for (int c0 = 0; c0 < nc; c0 += 16) {
            const int c = c0 + sub;
            if (c < nc) {
                const uint4* dp = reinterpret_cast<const uint4*>(V.node_desc + (size_t)V.children[off + c] * 32);
                const uint4 d0 = dp[0], d1 = dp[1];
                const uint32_t dist = __popc(f0.x ^ d0.x) + __popc(f0.y ^ d0.y) + __popc(f0.z ^ d0.z) + __popc(f0.w ^ d0.w) +
                                      __popc(f1.x ^ d1.x) + __popc(f1.y ^ d1.y) + __popc(f1.z ^ d1.z) + __popc(f1.w ^ d1.w);
                best = min(best, (dist << 22) | (uint32_t)c);      // distance <= 256, child position < 2^22
            }
        }
        best = row16_min_u32(best);
        node = (uint32_t)V.children[off + (int)(best & 0x3FFFFFu)];
        if (level == nid_level) { nid = node; nid_set = true; }
        off = V.child_offset[node]; nc = V.child_offset[node + 1] - off;
    }
    if (sub == 0) {
        if (!nid_set) nid = node;
        const double w = V.node_weight[node];
        const bool kept = w > 0;
        word_id[o] = kept ? V.node_word[node] : 0xFFFFFFFFu;
        node_id[o] = kept ? nid : 0xFFFFFFFFu;
        weight[o] = w;
    }
}

// ascending bitonic sort of N (power of two) 64-bit keys in LDS by 256 threads
__device__ void bitonic_sort_u64(unsigned long long* s, int N) {
    for (int k = 2; k <= N; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < N / 2; t += 256) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                const unsigned long long a = s[i], c = s[l];
                if ((a > c) == ((i & k) == 0)) { s[i] = c; s[l] = a; }
            }
            wg_barrier();
        }
}

// exclusive scan of one int per thread over the 256-thread block; returns the prefix, *total = block sum
__device__ int block_scan256(int v, int* s_part, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int u = __shfl_up(inc, d, 64);
        if (lane >= d) inc += u;
    }
    if (lane == 63) s_part[wave] = inc;
    wg_barrier();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += s_part[w];
    *total = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    wg_barrier();
    return base + inc - v;
}

__global__ __launch_bounds__(256) void k_bow_assemble(const uint32_t* __restrict__ word_id, const uint32_t* __restrict__ node_id,
                                                      const double* __restrict__ weight, const int32_t* __restrict__ counts, int cap, int N,
                                                      int accumulate, int norm, uint32_t* __restrict__ bow_word, double* __restrict__ bow_value,
                                                      int32_t* __restrict__ n_bow, uint32_t* __restrict__ fv_node, uint32_t* __restrict__ fv_feat,
                                                      int32_t* __restrict__ n_fv) {
    extern __shared__ unsigned long long s_dyn[];
    unsigned long long* keys = s_dyn;                                   // N
    double* vals = reinterpret_cast<double*>(s_dyn + N);                // N
    __shared__ int s_part[4];
    __shared__ int s_m;
    __shared__ double s_norm;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = counts ? min(counts[b], cap) : cap;
    const size_t base = (size_t)b * cap;
    const unsigned long long kNone = ~0ull;

    for (int pass = 0; pass < 2; ++pass) {
        const uint32_t* hi = pass == 0 ? word_id : node_id;
        for (int i = tid; i < N; i += 256) {
            unsigned long long k = kNone;
            if (i < n && word_id[base + i] != 0xFFFFFFFFu) k = ((unsigned long long)hi[base + i] << 32) | (unsigned)i;
            keys[i] = k;
        }
        if (tid == 0) s_m = 0;
        wg_barrier();
        bitonic_sort_u64(keys, N);
        for (int j = tid; j < N; j += 256)
            if (keys[j] != kNone && (j + 1 == N || keys[j + 1] == kNone)) s_m = j + 1;
        wg_barrier();
        const int m = s_m;
        if (pass == 1) {
            for (int j = tid; j < m; j += 256) { fv_node[base + j] = (uint32_t)(keys[j] >> 32); fv_feat[base + j] = (uint32_t)keys[j]; }
            if (tid == 0) n_fv[b] = m;
            break;
        }
        // distinct words: thread t owns the sorted entries [t * per, (t + 1) * per)
        const int per = N / 256 > 0 ? N / 256 : 1;
        const int j0 = tid * per, j1 = min(j0 + per, m);
        int heads = 0;
        for (int j = j0; j < j1; ++j) heads += (j == 0 || (uint32_t)(keys[j - 1] >> 32) != (uint32_t)(keys[j] >> 32));
        int nu;
        int pos = block_scan256(heads, s_part, &nu);
        for (int j = j0; j < j1; ++j) {
            const uint32_t w_id = (uint32_t)(keys[j] >> 32);
            if (j != 0 && (uint32_t)(keys[j - 1] >> 32) == w_id) continue;
            const double w = weight[base + (uint32_t)keys[j]];
            double v = w;
            if (accumulate)
                for (int t = j + 1; t < m && (uint32_t)(keys[t] >> 32) == w_id; ++t) v += w;
            vals[pos] = v;
            bow_word[base + pos] = w_id;
            ++pos;
        }
        wg_barrier();
        if (accumulate && norm == 0 && nu > 0) {
            const double nd = (double)nu;
            for (int p = tid; p < nu; p += 256) vals[p] /= nd;
            wg_barrier();
        }
        if (norm != 0) {
            if (tid == 0) {
                double s = 0.0;
                if (norm == 1) { for (int p = 0; p < nu; ++p) s += fabs(vals[p]); }
                else { for (int p = 0; p < nu; ++p) s += vals[p] * vals[p]; s = sqrt(s); }
                s_norm = s;
            }
            wg_barrier();
            const double s = s_norm;
            if (s > 0.0)
                for (int p = tid; p < nu; p += 256) vals[p] /= s;
            wg_barrier();
        }
        for (int p = tid; p < nu; p += 256) bow_value[base + p] = vals[p];
        if (tid == 0) n_bow[b] = nu;
        wg_barrier();
    }
}

}  // namespace
}  // namespace plp

using namespace plp;

struct plp_bow_vocab {
    int device = 0;
    hipStream_t stream = nullptr;
    int n_nodes = 0, L = 0, accumulate = 1, norm = 1;
    DevBuf child_offset, children, node_desc, node_weight, node_word;
    DevBuf word, node, weight, stage;          // per-feature scratch of the transform; slab of the host-pointer path
    std::mutex mu;
    BowTreeDev tree() const {
        return BowTreeDev{(const int32_t*)child_offset.p, (const int32_t*)children.p, (const uint8_t*)node_desc.p, (const double*)node_weight.p,
                          (const uint32_t*)node_word.p, L};
    }
};

extern "C" {

plp_status plp_bow_vocab_create(int device, const plp_bow_tree* t, plp_bow_vocab** out) {
    if (!t || !out) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    *out = nullptr;
    if (t->n_nodes < 2 || !t->child_offset || !t->children || !t->node_desc || !t->node_weight || !t->node_word)
        return set_error(PLP_ERR_INVALID_ARG, "the vocabulary is empty");
    if (t->L < 1 || t->L > 64 || (t->accumulate != 0 && t->accumulate != 1) || t->norm < 0 || t->norm > 2)
        return set_error(PLP_ERR_INVALID_ARG, "bad depth / weighting / norm");
    // a tree: offsets monotone, the root has children, every other node is the child of exactly one node
    const int n = t->n_nodes;
    if (t->child_offset[0] != 0) return set_error(PLP_ERR_INVALID_ARG, "child_offset[0] must be 0");
    for (int i = 0; i < n; ++i)
        if (t->child_offset[i + 1] < t->child_offset[i] || t->child_offset[i + 1] - t->child_offset[i] >= (1 << 22))
            return set_error(PLP_ERR_INVALID_ARG, "child_offset must not decrease (and a node has fewer than 2^22 children)");
    const int n_edges = t->child_offset[n];
    if (n_edges != n - 1 || t->child_offset[1] == 0) return set_error(PLP_ERR_INVALID_ARG, "not a tree rooted at node 0");
    std::vector<uint8_t> seen((size_t)n, 0);
    for (int e = 0; e < n_edges; ++e) {
        const int c = t->children[e];
        if (c <= 0 || c >= n || seen[(size_t)c]) return set_error(PLP_ERR_INVALID_ARG, "not a tree rooted at node 0");
        seen[(size_t)c] = 1;
    }
    // n - 1 edges, every non-root node with exactly one parent: still a forest plus cycles unless all are reachable
    {
        std::vector<int> stack{0};
        size_t reached = 0;
        while (!stack.empty()) {
            const int v = stack.back(); stack.pop_back(); ++reached;
            for (int e = t->child_offset[v]; e < t->child_offset[v + 1]; ++e) stack.push_back(t->children[e]);
        }
        if (reached != (size_t)n) return set_error(PLP_ERR_INVALID_ARG, "not a tree rooted at node 0");
    }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev) return set_error(PLP_ERR_NO_DEVICE, "no such HIP device");
    auto* v = new plp_bow_vocab();
    v->device = device; v->n_nodes = n; v->L = t->L; v->accumulate = t->accumulate; v->norm = t->norm;
    auto fail = [&](hipError_t e, const char* what) { delete v; return set_hip_error(e, what, __FILE__, __LINE__); };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return fail(e, "hipSetDevice");
    if ((e = hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
    if ((e = v->child_offset.upload(t->child_offset, sizeof(int32_t) * (size_t)(n + 1), v->stream)) != hipSuccess) return fail(e, "upload");
    if ((e = v->children.upload(t->children, sizeof(int32_t) * (size_t)n_edges, v->stream)) != hipSuccess) return fail(e, "upload");
    if ((e = v->node_desc.upload(t->node_desc, (size_t)n * 32, v->stream)) != hipSuccess) return fail(e, "upload");
    if ((e = v->node_weight.upload(t->node_weight, sizeof(double) * (size_t)n, v->stream)) != hipSuccess) return fail(e, "upload");
    if ((e = v->node_word.upload(t->node_word, sizeof(uint32_t) * (size_t)n, v->stream)) != hipSuccess) return fail(e, "upload");
    if ((e = hipStreamSynchronize(v->stream)) != hipSuccess) return fail(e, "hipStreamSynchronize");
    *out = v;
    return PLP_OK;
}

void plp_bow_vocab_destroy(plp_bow_vocab* v) {
    if (!v) return;
    (void)hipSetDevice(v->device);
    if (v->stream) { (void)hipStreamSynchronize(v->stream); (void)hipStreamDestroy(v->stream); }
    delete v;
}

// the transform proper; the caller holds v->mu (the scratch arrays word / node / weight / stage belong to the handle)
static plp_status bow_transform_locked(plp_bow_vocab* v, const uint8_t* d_desc, const int32_t* d_counts, int32_t cap, int32_t B, int32_t levelsup,
                                       uint32_t* d_word_id, uint32_t* d_node_id, uint32_t* d_bow_word, double* d_bow_value, int32_t* d_n_bow,
                                       uint32_t* d_fv_node, uint32_t* d_fv_feat, int32_t* d_n_fv, void* hip_stream) {
    if (!v || !d_desc || !d_bow_word || !d_bow_value || !d_n_bow || !d_fv_node || !d_fv_feat || !d_n_fv)
        return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    if (cap <= 0 || B <= 0 || levelsup < 0) return set_error(PLP_ERR_INVALID_ARG, "cap, B must be positive, levelsup >= 0");
    if (cap > 8192) return set_error(PLP_ERR_UNSUPPORTED, "more than 8192 descriptors per frame");   // (the per-frame maps are sorted in LDS: 16 bytes per descriptor, 128 KB of a CU's 160)
    PLP_HIP(hipSetDevice(v->device));
    const size_t tot = (size_t)B * cap;
    if (!d_word_id) { PLP_HIP(v->word.reserve(tot * 4)); d_word_id = (uint32_t*)v->word.p; }
    if (!d_node_id) { PLP_HIP(v->node.reserve(tot * 4)); d_node_id = (uint32_t*)v->node.p; }
    PLP_HIP(v->weight.reserve(tot * 8));
    hipStream_t st = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(k_bow_descend, dim3((cap + 15) / 16, B), dim3(256), 0, st, v->tree(), d_desc, d_counts, cap, levelsup, d_word_id, d_node_id,
                       (double*)v->weight.p);
    int N = 256;
    while (N < cap) N <<= 1;
    const size_t lds = (size_t)N * 16;
    if (lds > 48 * 1024) PLP_HIP(hipFuncSetAttribute((const void*)k_bow_assemble, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 16));
    hipLaunchKernelGGL(k_bow_assemble, dim3(B), dim3(256), lds, st, d_word_id, d_node_id, (const double*)v->weight.p, d_counts, cap, N, v->accumulate,
                       v->norm, d_bow_word, d_bow_value, d_n_bow, d_fv_node, d_fv_feat, d_n_fv);
    PLP_HIP(hipGetLastError());
    return PLP_OK;
}

plp_status plp_bow_transform_device(plp_bow_vocab* v, const uint8_t* d_desc, const int32_t* d_counts, int32_t cap, int32_t B, int32_t levelsup,
                                    uint32_t* d_word_id, uint32_t* d_node_id, uint32_t* d_bow_word, double* d_bow_value, int32_t* d_n_bow,
                                    uint32_t* d_fv_node, uint32_t* d_fv_feat, int32_t* d_n_fv, void* hip_stream) {
    if (!v) return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    std::lock_guard<std::mutex> lk(v->mu);
    return bow_transform_locked(v, d_desc, d_counts, cap, B, levelsup, d_word_id, d_node_id, d_bow_word, d_bow_value, d_n_bow, d_fv_node, d_fv_feat,
                                d_n_fv, hip_stream);
}

plp_status plp_bow_transform_host(plp_bow_vocab* v, const uint8_t* desc, int32_t n, int32_t levelsup, uint32_t* word_id, uint32_t* node_id,
                                  uint32_t* bow_word, double* bow_value, int32_t* n_bow, uint32_t* fv_node, uint32_t* fv_feat, int32_t* n_fv) {
    if (!v || !n_bow || !n_fv || n < 0 || (n > 0 && (!desc || !bow_word || !bow_value || !fv_node || !fv_feat)))
        return set_error(PLP_ERR_INVALID_ARG, "NULL argument");
    *n_bow = 0; *n_fv = 0;
    if (n == 0) return PLP_OK;
    if (n > 8192) return set_error(PLP_ERR_UNSUPPORTED, "more than 8192 descriptors per frame");
    uint8_t* slab;
    const size_t cap = (size_t)n;
    // desc | word | node | bow_word | bow_value | fv_node | fv_feat | n_bow, n_fv
    const size_t o_desc = 0, o_word = o_desc + cap * 32, o_node = o_word + cap * 4, o_bw = o_node + cap * 4, o_bv = (o_bw + cap * 4 + 7) & ~(size_t)7,
                 o_fn = o_bv + cap * 8, o_ff = o_fn + cap * 4, o_cnt = o_ff + cap * 4, total = o_cnt + 8;
    // one lock across staging, transform and read-back: DBoW2's transform is const and the reference calls one vocabulary
    // from several threads (frame::compute_bow, keyframe::compute_bow), so callers may share a handle
    std::lock_guard<std::mutex> lk(v->mu);
    PLP_HIP(hipSetDevice(v->device));
    PLP_HIP(v->stage.reserve(total));
    slab = (uint8_t*)v->stage.p;
    PLP_HIP(hipMemcpyAsync(slab + o_desc, desc, cap * 32, hipMemcpyHostToDevice, v->stream));
    PLP_TRY(bow_transform_locked(v, slab + o_desc, nullptr, n, 1, levelsup, (uint32_t*)(slab + o_word), (uint32_t*)(slab + o_node),
                                     (uint32_t*)(slab + o_bw), (double*)(slab + o_bv), (int32_t*)(slab + o_cnt), (uint32_t*)(slab + o_fn),
                                     (uint32_t*)(slab + o_ff), (int32_t*)(slab + o_cnt + 4), v->stream));
    int32_t cnt[2];
    PLP_HIP(hipMemcpyAsync(cnt, slab + o_cnt, 8, hipMemcpyDeviceToHost, v->stream));
    if (word_id) PLP_HIP(hipMemcpyAsync(word_id, slab + o_word, cap * 4, hipMemcpyDeviceToHost, v->stream));
    if (node_id) PLP_HIP(hipMemcpyAsync(node_id, slab + o_node, cap * 4, hipMemcpyDeviceToHost, v->stream));
    PLP_HIP(hipStreamSynchronize(v->stream));
    if (cnt[0] > 0) {
        PLP_HIP(hipMemcpyAsync(bow_word, slab + o_bw, (size_t)cnt[0] * 4, hipMemcpyDeviceToHost, v->stream));
        PLP_HIP(hipMemcpyAsync(bow_value, slab + o_bv, (size_t)cnt[0] * 8, hipMemcpyDeviceToHost, v->stream));
    }
    if (cnt[1] > 0) {
        PLP_HIP(hipMemcpyAsync(fv_node, slab + o_fn, (size_t)cnt[1] * 4, hipMemcpyDeviceToHost, v->stream));
        PLP_HIP(hipMemcpyAsync(fv_feat, slab + o_ff, (size_t)cnt[1] * 4, hipMemcpyDeviceToHost, v->stream));
    }
    PLP_HIP(hipStreamSynchronize(v->stream));
    *n_bow = cnt[0]; *n_fv = cnt[1];
    return PLP_OK;
}

}  // extern "C"
