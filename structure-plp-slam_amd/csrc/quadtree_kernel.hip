// K4  key-point distribution (quadtree) on the GPU: one 256-thread workgroup per (level, frame).
//
// Restates distribute_keypoints_via_tree (reference src/PLPSLAM/feature/orb_extractor.cc:468-685)
// in the sort-based form proven equivalent on the host in quadtree_model.hpp:
//   1. gather the level's per-cell candidate lists in reference order (cell row, cell col, row-major)
//   2. one tree key per candidate (initial node + 2 bits per split), LSD radix sort by key
//   3. every node = a range of the sorted array; whole-list split passes and the "fullest first"
//      fill phase run data-parallel over node records with prefix sums giving the list order
//   4. per node: maximum score, ties to the smallest candidate index
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#include "orb_device.hpp"
#include "plp_barrier.hpp"
#include "plp_common.hpp"
#include "quadtree_model.hpp"

namespace plp {

constexpr int kQtMaxNodes = kQtMaxNodesLds;   // hard upper bound of the node arrays: what 160 KB of LDS hold (orb_device.hpp; 3 * quota + 8 nodes: quota up to 1960 per level)
// One block of LDS serves as the cell prefix (n_cells words), then the radix counters [16 digits][segments] (u16) of levels of up to blk_words / 8 wave-sized
// segments, then the sorted-key cache (blk_words keys); larger levels use the HBM scratch.  blk_words = 2048 (8 KB: 256 segments = 16384 candidates in LDS) -- or 1024
// for a per-level quota above 1960 (round 6: the 4 KB that buys fit the node arrays of one level at K = 2000, the reference's single-level configuration).
constexpr int kQtBlkWords = 2048, kQtBlkWordsSmall = 1024;

// LDS of one workgroup, carved from dynamic shared memory: the node arrays are sized by the largest per-level quota
// (a list never holds more than 3 * quota + 3 nodes), so that 2-3 workgroups fit a CU for the usual K = 1000..2000
// (a fixed 85 KB layout had left one workgroup = four waves per CU).
struct QtShared {
    uint16_t* cnt;                  // [16 * kQtSegLds] radix counters [digit][segment]      } one 8 KB block
    uint32_t* big;                  // [kQtKeyCache] cell prefix, later the sorted-key cache  }
    // node records, two generations (ping-pong): start / end of the node's range, depth, leaf flag.  Addressed by arithmetic:
    // arrays of pointers indexed by the generation had put the whole struct into scratch memory.
    uint16_t* ns0;                  // ns(0), ne(0), ns(1), ne(1): max_nodes entries each
    uint8_t* nd0;                   // nd(0), nleaf(0), nd(1), nleaf(1): max_nodes bytes each
    __device__ __forceinline__ uint16_t* ns(int k) const { return ns0 + (size_t)k * 2 * max_nodes; }
    __device__ __forceinline__ uint16_t* ne(int k) const { return ns0 + (size_t)k * 2 * max_nodes + max_nodes; }
    __device__ __forceinline__ uint8_t* nd(int k) const { return nd0 + (size_t)k * 2 * max_nodes; }
    __device__ __forceinline__ uint8_t* nleaf(int k) const { return nd0 + (size_t)k * 2 * max_nodes + max_nodes; }
    uint16_t *b1, *b2, *b3;
    uint32_t* pk;                   // packed per-entry counts -> exclusive prefixes (lo16 created, hi16 kept)
    uint16_t *pool_pos, *pool_sorted;
    uint32_t* partial;
    int* misc;
    int max_nodes;
};
__host__ __device__ inline size_t qt_lds_bytes(int max_nodes, int blk_words) {
    return 4 * (size_t)blk_words + (size_t)max_nodes * (4 * 2 + 4 * 1 + 3 * 2 + 4 + 2 * 2) + 16 + 32;
}
__device__ __forceinline__ QtShared qt_carve(uint8_t* base, int mn, int blk_words) {
    QtShared S;
    S.cnt = reinterpret_cast<uint16_t*>(base); S.big = reinterpret_cast<uint32_t*>(base);
    uint8_t* p = base + 4 * (size_t)blk_words;
    S.pk = reinterpret_cast<uint32_t*>(p); p += 4 * (size_t)mn;
    S.partial = reinterpret_cast<uint32_t*>(p); p += 16;
    S.misc = reinterpret_cast<int*>(p); p += 32;
    S.ns0 = reinterpret_cast<uint16_t*>(p); p += 4 * 2 * (size_t)mn;
    S.b1 = reinterpret_cast<uint16_t*>(p); p += 2 * (size_t)mn; S.b2 = reinterpret_cast<uint16_t*>(p); p += 2 * (size_t)mn;
    S.b3 = reinterpret_cast<uint16_t*>(p); p += 2 * (size_t)mn;
    S.pool_pos = reinterpret_cast<uint16_t*>(p); p += 2 * (size_t)mn; S.pool_sorted = reinterpret_cast<uint16_t*>(p); p += 2 * (size_t)mn;
    S.nd0 = p; p += 4 * (size_t)mn;
    S.max_nodes = mn;
    return S;
}

// in-place exclusive scan of a[0..M) by the whole workgroup; returns the total.  Per-thread chunk sums are scanned
// inside each wave with shuffles; only the four wave totals cross a barrier (the kernel runs ~50 of these scans per
// level, and a 16-barrier Hillis-Steele over 256 partials had made barriers its main cost).
// An array that lives in LDS or in HBM, decided per level (wave-uniform): every access is a DS or a GLOBAL instruction behind a scalar branch -- never a FLAT one (a
// pointer chosen at run time, `n <= cap ? lds : hbm`, had made every access FLAT: round 6) -- and the code around it exists once (inlining the radix
// passes and the division once per place cost 0.1 ms per step: 83 spilled SGPRs, twice the code; `tools/sessions/r06/run38.sh`).
template <typename T> struct DualArr {
    __attribute__((address_space(3))) T* l; T* g; bool in_lds;
    __device__ __forceinline__ T operator[](int i) const { T v; if (in_lds) v = l[i]; else v = g[i]; return v; }
    __device__ __forceinline__ void set(int i, T v) const { if (in_lds) l[i] = v; else g[i] = v; }
};
template <typename T> __device__ __forceinline__ DualArr<T> dual_arr(T* lds_ptr, T* hbm_ptr, bool in_lds) {
    return DualArr<T>{(__attribute__((address_space(3))) T*)lds_ptr, hbm_ptr, in_lds};
}
struct PtrArr32 { uint32_t* p; __device__ __forceinline__ uint32_t operator[](int i) const { return p[i]; } __device__ __forceinline__ void set(int i, uint32_t v) const { p[i] = v; } };

template <typename T, class Arr>
__device__ uint32_t block_scan_excl_arr(const Arr& a, int M, uint32_t* partial) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int chunk = (M + 255) >> 8;
    const int lo = min(tid * chunk, M), hi = min(lo + chunk, M);
    uint32_t sum = 0;
    for (int i = lo; i < hi; ++i) sum += a[i];
    uint32_t inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) partial[wv] = inc;
    wg_barrier();
    const uint32_t w0 = partial[0], w1 = partial[1], w2 = partial[2], w3 = partial[3];
    const uint32_t base = (wv > 0 ? w0 : 0) + (wv > 1 ? w1 : 0) + (wv > 2 ? w2 : 0);
    uint32_t run = base + inc - sum;
    for (int i = lo; i < hi; ++i) { const uint32_t v = a[i]; a.set(i, (T)run); run += v; }
    wg_barrier();
    return w0 + w1 + w2 + w3;
}

template <typename T>
__device__ uint32_t block_scan_excl(T* a, int M, uint32_t* partial) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int chunk = (M + 255) >> 8;
    const int lo = min(tid * chunk, M), hi = min(lo + chunk, M);
    uint32_t sum = 0;
    for (int i = lo; i < hi; ++i) sum += a[i];
    uint32_t inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) partial[wv] = inc;
    wg_barrier();
    const uint32_t w0 = partial[0], w1 = partial[1], w2 = partial[2], w3 = partial[3];
    const uint32_t base = (wv > 0 ? w0 : 0) + (wv > 1 ? w1 : 0) + (wv > 2 ? w2 : 0);
    uint32_t run = base + inc - sum;
    for (int i = lo; i < hi; ++i) { const uint32_t v = a[i]; a[i] = (T)run; run += v; }
    wg_barrier();
    return w0 + w1 + w2 + w3;
}

__device__ __forceinline__ uint32_t dev_qt_key(int x, int y, const LevelDev& L) {
    const unsigned ix = (unsigned)((double)(float)x / L.delta_x);
    const unsigned iy = (unsigned)((double)(float)y / L.delta_y);
    const unsigned node = ix + iy * (unsigned)L.n_init_x;
    int bx = (int)(L.delta_x * ix), ex = (int)(L.delta_x * (ix + 1));
    int by = (int)(L.delta_y * iy), ey = (int)(L.delta_y * (iy + 1));
    uint32_t key = node;
#pragma unroll
    for (int d = 0; d < kQtDepth; ++d) {
        const int mx = bx + ((ex - bx + 1) >> 1), my = by + ((ey - by + 1) >> 1);
        const unsigned xb = mx <= x, yb = my <= y;
        bx = xb ? mx : bx; ex = xb ? ex : mx;
        by = yb ? my : by; ey = yb ? ey : my;
        key = (key << 2) | (xb | (yb << 1));
    }
    return key;
}

// split node [s,e) at `depth`: boundaries of digit values 1,2,3; returns number of non-empty children
template <class KeyArr> __device__ __forceinline__ int dev_split(const KeyArr& K, int s, int e, int depth, int& o1, int& o2, int& o3) {
    if (depth >= kQtDepth) { o1 = o2 = o3 = e; return 1; }
    const int sh = 2 * (kQtDepth - 1 - depth);
    if (e - s == 1) {
        const unsigned d = (K[s] >> sh) & 3u;
        o1 = d >= 1 ? s : e; o2 = d >= 2 ? s : e; o3 = d >= 3 ? s : e;
        return 1;
    }
    int b[3];
#pragma unroll
    for (unsigned v = 1; v <= 3; ++v) {
        int lo = s, hi = e;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (((K[mid] >> sh) & 3u) >= v) hi = mid; else lo = mid + 1;
        }
        b[v - 1] = lo;
    }
    o1 = b[0]; o2 = b[1]; o3 = b[2];
    return (o1 > s) + (o2 > o1) + (o3 > o2) + (e > o3);
}

__global__ __launch_bounds__(256) void k_quadtree(const LevelDev* __restrict__ lv, int n_cells_total,
                                                  const uint32_t* __restrict__ cell_cand, const int32_t* __restrict__ cell_count,
                                                  int32_t* __restrict__ sel, int32_t* __restrict__ sel_count, int total_sel_cap,
                                                  uint8_t* __restrict__ scratch, size_t scratch_frame_stride,
                                                  int32_t* __restrict__ status, int max_nodes, int blk_words) {
    extern __shared__ __attribute__((aligned(16))) uint8_t qt_smem[];
    const QtShared S = qt_carve(qt_smem, max_nodes, blk_words);
    const int kQtSegLds = blk_words >> 3, kQtKeyCache = blk_words;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int level = blockIdx.x, frame = blockIdx.y;
    const LevelDev L = lv[level];
    const int N = L.quota;
    uint32_t* cand = reinterpret_cast<uint32_t*>(scratch + (size_t)frame * scratch_frame_stride + L.qt_off);
    uint32_t* key = cand + L.qt_cap;
    uint32_t* idxA = key + L.qt_cap;
    uint32_t* idxB = idxA + L.qt_cap;
    uint16_t* cnt_hbm = reinterpret_cast<uint16_t*>(idxB + L.qt_cap);   // radix counters of levels with more than 32768 candidates
    int32_t* out_sel = sel + (size_t)frame * total_sel_cap + L.sel_base;
    int32_t* out_cnt = sel_count + frame * kMaxLevels + level;

    // ---- 1. gather candidates in reference order + keys
    const int32_t* cc = cell_count + (size_t)frame * n_cells_total + L.cell_base;
    for (int c = tid; c < L.n_cells; c += 256) S.big[c] = (uint32_t)cc[c];
    wg_barrier();
    int n = (int)block_scan_excl(S.big, L.n_cells, S.partial);
    if (n > L.qt_cap) { if (tid == 0) atomicOr(status, 2); n = L.qt_cap; }
    for (int c = wv; c < L.n_cells; c += 4) {
        const int base = (int)S.big[c], cnt = cc[c];
        const uint32_t* src = cell_cand + ((size_t)frame * n_cells_total + L.cell_base + c) * kCellCap;
        for (int j = lane; j < cnt; j += 64) {
            const int dst = base + j;
            if (dst < n) {
                const uint32_t pk = src[j];
                cand[dst] = pk;
                key[dst] = dev_qt_key((int)(pk & 0xfff), (int)((pk >> 12) & 0xfff), L);
            }
        }
    }
    wg_barrier();
    if (n == 0) { if (tid == 0) *out_cnt = 0; return; }

    // ---- 2. LSD radix sort (4 bits per pass) of candidate indices by key bits [sort_lo, sort_hi)
    const int nseg = (n + 63) >> 6;
    uint32_t* src = idxA;
    uint32_t* dst = idxB;
    bool first = true;
    // The radix counters live in LDS, or -- levels with more than blk_words / 8 segments -- in HBM: a DualArr (DS or GLOBAL instructions behind a scalar branch)
    const DualArr<uint16_t> CNT = dual_arr(S.cnt, cnt_hbm, nseg <= kQtSegLds);
    for (int shift = L.sort_lo; shift < L.sort_hi; shift += 4) {
        for (int seg = wv; seg < nseg; seg += 4) {
            const int i = seg * 64 + lane;
            unsigned d = 16;
            if (i < n) d = (key[first ? (uint32_t)i : src[i]] >> shift) & 15u;
            uint32_t mine = 0;
#pragma unroll
            for (unsigned v = 0; v < 16; ++v) {
                const unsigned long long bal = __ballot(d == v);
                if ((unsigned)lane == v) mine = (uint32_t)__popcll(bal);
            }
            if (lane < 16) CNT.set(lane * nseg + seg, (uint16_t)mine);
        }
        wg_barrier();
        block_scan_excl_arr<uint16_t>(CNT, 16 * nseg, S.partial);
        for (int seg = wv; seg < nseg; seg += 4) {
            const int i = seg * 64 + lane;
            unsigned d = 16;
            uint32_t id = 0;
            if (i < n) { id = first ? (uint32_t)i : src[i]; d = (key[id] >> shift) & 15u; }
            // the lanes holding the same digit as this one (d = 16: the lanes past the end), from one ballot per BIT of d; selecting the
            // lane's own of sixteen per-value ballots had cost 30 registers for their copies (the kernel's register peak)
            unsigned long long mybal = ~0ull;
#pragma unroll
            for (unsigned bit = 0; bit < 5; ++bit) {
                const bool one = (d >> bit) & 1u;
                const unsigned long long bal = __ballot(one);
                mybal &= one ? bal : ~bal;
            }
            if (i < n) dst[(uint32_t)CNT[d * nseg + seg] + (uint32_t)__popcll(mybal & ((1ull << lane) - 1ull))] = id;
        }
        wg_barrier();
        uint32_t* t = src; src = dst; dst = t;
        first = false;
    }
    // src now holds candidate indices in key order (identity if no pass ran); dst <- sorted keys
    for (int i = tid; i < n; i += 256) {
        const uint32_t id = first ? (uint32_t)i : src[i];
        if (first) src[i] = id;
        const uint32_t k = key[id];
        dst[i] = k;
        if (n <= kQtKeyCache) S.big[i] = k;
    }
    wg_barrier();
    const uint32_t* sidx = src;
    // The sorted keys are read from their LDS copy when it exists, from the HBM scratch otherwise: a DualArr again
    const DualArr<uint32_t> K = dual_arr(S.big, dst, n <= kQtKeyCache);

    // ---- 3a. initial nodes = runs of equal initial-node index
    int cur = 0;
    if (tid == 0) {
        int m = 0, s = 0;
        while (s < n) {   // at most n_init (<= 8) runs; binary search each end
            const uint32_t node = K[s] >> (2 * kQtDepth);
            int lo = s, hi = n;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if ((K[mid] >> (2 * kQtDepth)) > node) hi = mid; else lo = mid + 1; }
            S.ns(0)[m] = (uint16_t)s; S.ne(0)[m] = (uint16_t)lo; S.nd(0)[m] = 0; S.nleaf(0)[m] = (lo - s == 1);
            ++m; s = lo;
        }
        S.misc[0] = m;
    }
    wg_barrier();
    int m = S.misc[0];
    bool filled = false, failed = false;

    // ---- 3b. phase 1: split every non-leaf entry, children to the front in reverse creation order
    while (true) {
        const int prev = m;
        if (tid == 0) S.misc[1] = 0;   // pool size (children with more than one point)
        wg_barrier();
        for (int i = tid; i < m; i += 256) {
            if (S.nleaf(cur)[i]) { S.pk[i] = 1u << 16; continue; }
            int o1, o2, o3;
            const int s = S.ns(cur)[i], e = S.ne(cur)[i];
            const int k = dev_split(K, s, e, S.nd(cur)[i], o1, o2, o3);
            S.b1[i] = (uint16_t)o1; S.b2[i] = (uint16_t)o2; S.b3[i] = (uint16_t)o3;
            S.pk[i] = (uint32_t)k;
            const int big = (o1 - s > 1) + (o2 - o1 > 1) + (o3 - o2 > 1) + (e - o3 > 1);
            if (big) atomicAdd(&S.misc[1], big);
        }
        wg_barrier();
        const uint32_t tot = block_scan_excl(S.pk, m, S.partial);
        const int T = (int)(tot & 0xffff), kept = (int)(tot >> 16);
        const int m_new = T + kept;
        if (m_new > S.max_nodes) { failed = true; break; }
        const int nxt = cur ^ 1;
        for (int i = tid; i < m; i += 256) {
            const uint32_t p = S.pk[i];
            const int s = S.ns(cur)[i], e = S.ne(cur)[i];
            if (S.nleaf(cur)[i]) {
                const int pos = T + (int)(p >> 16);
                S.ns(nxt)[pos] = (uint16_t)s; S.ne(nxt)[pos] = (uint16_t)e; S.nd(nxt)[pos] = S.nd(cur)[i]; S.nleaf(nxt)[pos] = 1;
                continue;
            }
            int q = (int)(p & 0xffff);
            const int b[5] = {s, S.b1[i], S.b2[i], S.b3[i], e};
            const uint8_t dep = (uint8_t)min((int)S.nd(cur)[i] + 1, 255);
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (b[c + 1] > b[c]) {
                    const int pos = T - 1 - q;
                    S.ns(nxt)[pos] = (uint16_t)b[c]; S.ne(nxt)[pos] = (uint16_t)b[c + 1]; S.nd(nxt)[pos] = dep; S.nleaf(nxt)[pos] = 0;
                    ++q;
                }
        }
        wg_barrier();
        cur = nxt; m = m_new;
        const int pool = S.misc[1];
        if (N <= m || m == prev) { filled = true; break; }
        if (N < m + pool) break;
        wg_barrier();
    }

    // ---- 3c. phase 2: split the fullest entries first until the quota is reached
    while (!filled && !failed) {
        const int prev = m;
        // pool = entries with more than one point, in list order (ordered compaction)
        for (int i = tid; i < m; i += 256) S.pk[i] = (uint32_t)(S.ne(cur)[i] - S.ns(cur)[i] > 1);
        wg_barrier();
        const int p = (int)block_scan_excl(S.pk, m, S.partial);
        for (int i = tid; i < m; i += 256)
            if (S.ne(cur)[i] - S.ns(cur)[i] > 1) S.pool_pos[S.pk[i]] = (uint16_t)i;
        wg_barrier();
        if (p == 0) break;   // nothing left to split
        // order: count descending, then list position ascending (== creation descending)
        for (int a = tid; a < p; a += 256) {
            const int ia = S.pool_pos[a];
            const uint32_t ka = ((uint32_t)(S.ne(cur)[ia] - S.ns(cur)[ia]) << 12) | (uint32_t)(4095 - min(ia, 4095));
            int rank = 0;
            for (int b = 0; b < p; ++b) {
                const int ib = S.pool_pos[b];
                const uint32_t kb = ((uint32_t)(S.ne(cur)[ib] - S.ns(cur)[ib]) << 12) | (uint32_t)(4095 - min(ib, 4095));
                rank += kb > ka;
            }
            S.pool_sorted[rank] = (uint16_t)ia;
        }
        wg_barrier();
        // split in sorted order; inclusive gain prefix decides where the quota is met
        for (int j = tid; j < p; j += 256) {
            const int i = S.pool_sorted[j];
            int o1, o2, o3;
            const int k = dev_split(K, S.ns(cur)[i], S.ne(cur)[i], S.nd(cur)[i], o1, o2, o3);
            S.b1[j] = (uint16_t)o1; S.b2[j] = (uint16_t)o2; S.b3[j] = (uint16_t)o3;
            S.pk[j] = (uint32_t)k;
        }
        if (tid == 0) S.misc[2] = p;   // r+1 = number of pool entries actually split
        wg_barrier();
        block_scan_excl(S.pk, p, S.partial);   // exclusive prefix of k over sorted order
        for (int j = tid; j < p; j += 256) {
            const int i = S.pool_sorted[j];
            int k = 1;
            {   // recover k_j from the boundaries
                const int s = S.ns(cur)[i], e = S.ne(cur)[i];
                k = (S.b1[j] > s) + (S.b2[j] > S.b1[j]) + (S.b3[j] > S.b2[j]) + (e > S.b3[j]);
            }
            const int size_after = m + (int)S.pk[j] + k - (j + 1);   // m + sum_{t<=j} (k_t - 1)
            if (N <= size_after) atomicMin(&S.misc[2], j + 1);
        }
        wg_barrier();
        const int nsplit = S.misc[2];
        // total children created by the first nsplit entries
        int T;
        {
            const int jl = nsplit - 1, il = S.pool_sorted[jl];
            const int s = S.ns(cur)[il], e = S.ne(cur)[il];
            const int kl = (S.b1[jl] > s) + (S.b2[jl] > S.b1[jl]) + (S.b3[jl] > S.b2[jl]) + (e > S.b3[jl]);
            T = (int)S.pk[jl] + kl;
        }
        const int m_new = T + m - nsplit;
        if (m_new > S.max_nodes) { failed = true; break; }
        const int nxt = cur ^ 1;
        // erased flags + survivors' new positions
        wg_barrier();
        for (int j = tid; j < nsplit; j += 256) {
            const int i = S.pool_sorted[j];
            int q = (int)S.pk[j];
            const int s = S.ns(cur)[i], e = S.ne(cur)[i];
            const int b[5] = {s, S.b1[j], S.b2[j], S.b3[j], e};
            const uint8_t dep = (uint8_t)min((int)S.nd(cur)[i] + 1, 255);
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (b[c + 1] > b[c]) {
                    const int pos = T - 1 - q;
                    S.ns(nxt)[pos] = (uint16_t)b[c]; S.ne(nxt)[pos] = (uint16_t)b[c + 1]; S.nd(nxt)[pos] = dep; S.nleaf(nxt)[pos] = 0;
                    ++q;
                }
            S.nleaf(cur)[i] |= 2;   // mark erased
        }
        wg_barrier();
        for (int i = tid; i < m; i += 256) S.pk[i] = (S.nleaf(cur)[i] & 2) ? 0u : 1u;
        wg_barrier();
        block_scan_excl(S.pk, m, S.partial);
        for (int i = tid; i < m; i += 256)
            if (!(S.nleaf(cur)[i] & 2)) {
                const int pos = T + (int)S.pk[i];
                S.ns(nxt)[pos] = S.ns(cur)[i]; S.ne(nxt)[pos] = S.ne(cur)[i]; S.nd(nxt)[pos] = S.nd(cur)[i]; S.nleaf(nxt)[pos] = S.nleaf(cur)[i];
            }
        wg_barrier();
        cur = nxt; m = m_new;
        if (N <= m || m == prev) break;
    }

    if (failed) {
        if (tid == 0) { atomicOr(status, 2); *out_cnt = 0; }
        return;
    }
    // ---- 4. best candidate per node: max score, then smallest candidate index
    const int m_out = min(m, L.sel_cap);
    if (m > L.sel_cap && tid == 0) atomicOr(status, 2);
    for (int i = tid; i < m_out; i += 256) {
        const int s = S.ns(cur)[i], e = S.ne(cur)[i];
        if (s >= n || e > n || s >= e) { atomicOr(status, 8); out_sel[i] = 0; continue; }   // cannot happen: node ranges partition [0, n)
        uint32_t best_id = sidx[s];
        if (best_id >= (uint32_t)n) { atomicOr(status, 8); best_id = (uint32_t)(n - 1); }   // cannot happen: the sort permutes [0, n)
        uint32_t best_pk = cand[best_id];
        for (int t = s + 1; t < e; ++t) {
            uint32_t id = sidx[t];
            if (id >= (uint32_t)n) { atomicOr(status, 8); id = (uint32_t)(n - 1); }
            const uint32_t pk = cand[id];
            const uint32_t sa = pk >> 24, sb = best_pk >> 24;
            if (sa > sb || (sa == sb && id < best_id)) { best_id = id; best_pk = pk; }
        }
        out_sel[i] = (int32_t)best_pk;
    }
    if (tid == 0) *out_cnt = m_out;
}

size_t quadtree_scratch_bytes_per_frame(const LevelDev* h_lv, int n_levels) {
    size_t end = 0;
    for (int l = 0; l < n_levels; ++l) end = h_lv[l].qt_off + (size_t)h_lv[l].qt_cap * 20;
    return (end + 255) / 256 * 256;
}

void launch_quadtree(hipStream_t st, const LevelDev* d_lv, int n_levels, int n_cells_total, const uint32_t* cell_cand,
                     const int32_t* cell_count, int32_t* sel, int32_t* sel_count, int total_sel_cap, uint32_t* qt_scratch,
                     size_t qt_scratch_frame_stride, int32_t* status, int B, int max_quota) {
    // node arrays for 3 * quota + 8 entries (the bound of the lists, see QtShared), rounded to 8: at K = 1000 (quota 217) 25.5 KB, so that THREE
    // workgroups fit the 77 KB two region-growing workgroups leave on a CU (rounded to a power of two it was 34.9 KB: two)
    const int want = std::max(256, (3 * max_quota + 8 + 7) / 8 * 8);
    const bool small_blk = want > kQtMaxNodes;                    // a quota above 1960 (plp_orb_create has checked that it fits with the smaller block and that no level has more cells than it holds)
    const int mn = std::min(small_blk ? kQtMaxNodesLdsSmallBlock : kQtMaxNodes, want), blk = small_blk ? kQtBlkWordsSmall : kQtBlkWords;
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_quadtree), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
    hipLaunchKernelGGL(k_quadtree, dim3(n_levels, B), dim3(256), qt_lds_bytes(mn, blk), st, d_lv, n_cells_total, cell_cand, cell_count, sel,
                       sel_count, total_sel_cap, reinterpret_cast<uint8_t*>(qt_scratch), qt_scratch_frame_stride, status, mn, blk);
}

}  // namespace plp
