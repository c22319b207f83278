// LSD seed order exactly as a reference built with libstdc++ produces it (closes definition D1; seed_sort_model.hpp has the
// derivation and the host model).  lsd.cpp's ll_angle sorts ALL (sw - 1)(sh - 1) pixels by gradient bin with std::sort; the kernel
// replays std::__introsort_loop on that array as rank-paired partitions, the stable counting sort of k_lsd_order_entries is the final
// insertion sort.
//
// One workgroup of 16 waves per frame, the array in HBM / L2 (`ent`, 4 bytes per pixel: pixel | defined << 19 | bin << 20):
//   G  segments longer than the LDS window (kSsT entries): the whole workgroup partitions one segment in global memory
//      (wg_partition<true>: chunk masks -> prefix sums -> partner positions by rank -> swaps; four barriers)
//   W  a segment that fits is loaded into LDS and finished there:
//      - above kSsTask entries the whole workgroup partitions it (wg_partition<false>, the same code on LDS)
//      - 65 .. kSsTask entries: ONE WAVE per segment, chunk masks in registers (lane c = chunk c), a task queue in LDS, no barriers
//      - 17 .. 64 entries: one LANE per segment runs the library's sequential loop
//      - at most 16 entries: left to the final insertion sort (= the counting sort)
//   The recursion budget travels with every segment; where it runs out one lane runs the library's heap sort (never seen on an
//   image; forced in the tests).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "line_device.hpp"
#include "seed_sort_model.hpp"

namespace plp {

namespace {

using seedsort::KeyDesc;
using seedsort::median3_pos;
using seedsort::select64;
using seedsort::select64_top;

constexpr int kSsWaves = 16, kSsThreads = 64 * kSsWaves;
#ifdef PLP_SS_RPOS32
constexpr int kSsT = 16384;
typedef uint32_t lrpos_t;
#else
constexpr int kSsT = 24576;          // LDS window, entries
typedef uint16_t lrpos_t;
#endif
constexpr int kSsTask = 4096;        // longest segment one wave partitions (64 chunks = 64 lanes)
constexpr int kSsSmall = 64;         // longest segment one lane sorts
constexpr int kSsShift = seedsort::kKeyShift;
static_assert(kSsShift == kLsdSeedPixBits, "entry layout = the seed packing of k_lsd_order");
constexpr uint32_t kSsDefBit = 1u << 19;
static_assert(kLsdMaxScaledPixels <= (size_t)kSsDefBit, "pixel index below the defined bit");
constexpr int kSsQueue = 512, kSsSmallCap = 1536, kSsGStack = 64, kSsWStack = 8;
static_assert(kSsT / (kSsSmall + 1) < kSsQueue && kSsT / 17 < kSsSmallCap && kSsT / (kSsTask + 1) < kSsWStack, "queue capacities");
constexpr int kSsChunksW = kSsT / 64 + 1;                                  // chunk slots of an LDS partition
constexpr int kSsMaxChunksG = (int)((kLsdMaxScaledPixels + 63) / 64) + 1;  // of a global one (prefix arrays alias the window)
static_assert(kSsMaxChunksG <= 8 * kSsThreads, "a thread scans at most 8 chunks (wg_partition)");
static_assert((size_t)kSsMaxChunksG * 12 <= (size_t)kSsT * 4, "G prefix arrays fit the idle window");

struct SsCtl {                       // LDS control block
    int bc[8];                       // broadcasts: m, cut, block-scan totals
    int wsum[2][kSsWaves];
    int g_n; int g_first[kSsGStack], g_last[kSsGStack], g_depth[kSsGStack];
    int w_n; int w_first[kSsWStack], w_last[kSsWStack], w_depth[kSsWStack];
    int q_head, q_tail, q_pending, q_open, n_small, overflow;
    long long t[24];                      // diagnostics (debug entry only): cycles in {G partitions, window load, workgroup levels, wave tasks, lanes, store}, windows, G partitions
};

// dynamic LDS layout (bytes)
constexpr size_t kOffWin = 0;
constexpr size_t kOffRpos = kOffWin + (size_t)kSsT * 4;
constexpr size_t kOffML = kOffRpos + ((size_t)(kSsT / 2 + 64) * sizeof(lrpos_t) + 7) / 8 * 8;
constexpr size_t kOffMR = kOffML + (size_t)kSsChunksW * 8;
constexpr size_t kOffPA = kOffMR + (size_t)kSsChunksW * 8;
constexpr size_t kOffPBx = kOffPA + (size_t)kSsChunksW * 4;
constexpr size_t kOffPBi = kOffPBx + (size_t)kSsChunksW * 4;
constexpr size_t kOffQ = (kOffPBi + (size_t)kSsChunksW * 4 + 7) / 8 * 8;
constexpr size_t kOffSmall = kOffQ + (size_t)kSsQueue * 8;
constexpr size_t kOffCtl = kOffSmall + (size_t)kSsSmallCap * 4;
constexpr size_t kSsLdsBytes = kOffCtl + sizeof(SsCtl);
static_assert(kSsLdsBytes <= 160 * 1024, "LDS of the seed sort");

struct SsMem {
    uint32_t* a;                      // the entries the partition works on (global: the frame's array; LDS: the window)
    void* rpos;                       // partner positions by rank: uint32_t[] (global) / uint16_t[] (LDS)
    unsigned long long *mL, *mR;      // chunk masks of left / right stoppers
    uint32_t *pA, *pBx, *pBi;         // per chunk: left stoppers before it / right stoppers after it / in it and after it (LDS in both cases)
};

// inclusive scan over the 64 lanes: rows of 16 by DPP row shifts (a lane whose source falls outside its row keeps the 0 passed as `old`), the
// three row totals through v_readlane -- no LDS round trips (six ds_bpermute steps had been a tenth of a small partition)
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47);
    return v + (lane >= 16 ? t0 : 0) + (lane >= 32 ? t1 : 0) + (lane >= 48 ? t2 : 0);
}
__device__ __forceinline__ int wave_max(int v) {   // the same value in every lane
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false));   // row_ror:8
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x122, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x121, 0xf, 0xf, false));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int src) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long lanes_below(int lane) { return (1ull << lane) - 1ull; }
__device__ __forceinline__ unsigned long long bits_above(unsigned long long m, int lane) { return (m >> lane) >> 1; }

// number of chunks c in [0, nch) with pred(c), for a predicate that is true on a prefix of the chunks: 64 probes per round (one wave)
template <class Pred> __device__ __forceinline__ int wave_count_prefix(int nch, int lane, Pred pred) {
    int lo = 0, hi = nch;              // the count lies in [lo, hi]
    while (lo < hi) {
        const int step = (hi - lo + 63) >> 6, c = lo + lane * step;
        const int t = __popcll(__ballot(c < hi && pred(c)));
        if (t == 0) { hi = lo; break; }
        const int nlo = lo + (t - 1) * step + 1;
        hi = min(hi, lo + t * step);
        lo = nlo;
    }
    return lo;
}

// Whole-workgroup partition of [first, last) (more than 64 entries) around the median of three; returns the cut.  Ends with a barrier.
// Every pass takes U chunks per wave and trip, so that U independent loads (and, in the swap pass, U chains of three dependent ones) are
// in flight per wave: with one chunk per trip the global-memory form ran at one memory latency per 64 entries and wave.
template <bool G> __device__ __forceinline__ int wg_partition(const SsMem& M, SsCtl* ctl, int first, int last, bool prof = false) {
    long long tp0 = prof ? clock64() : 0;
    auto plap = [&](int k) { if (prof && threadIdx.x == 0) { const long long t1 = clock64(); ctl->t[(G ? 16 : 10) + k] += t1 - tp0; tp0 = t1; } };
    using rpos_t = typename std::conditional<G, uint32_t, lrpos_t>::type;
    constexpr int U = G ? 8 : 4;
    rpos_t* rpos = (rpos_t*)M.rpos + (first >> 1);
    uint32_t* a = M.a;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int p0 = first + 1, np = last - p0, nch = (np + 63) >> 6, mid = first + (last - first) / 2;
    const uint32_t e0 = a[first], ea = a[p0], eb = a[mid], ec = a[last - 1];
    const int m3 = median3_pos(ea >> kSsShift, eb >> kSsShift, ec >> kSsShift, p0, mid, last - 1);
    const uint32_t em = m3 == p0 ? ea : (m3 == mid ? eb : ec), pk = em >> kSsShift;
    // A: stoppers of the array as it is after the median's swap with a[first] (a[m3] = e0; done physically in S)
    for (int c0 = wv * U; c0 < nch; c0 += kSsWaves * U) {
        uint32_t e[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pos = p0 + ((c0 + u) << 6) + lane;
            e[u] = pos < last ? a[pos] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u, pos = p0 + (c << 6) + lane;
            if (c >= nch) break;
            const bool valid = pos < last;
            const uint32_t k = (pos == m3 ? e0 : e[u]) >> kSsShift;
            const unsigned long long mL = __ballot(valid && k <= pk), mR = __ballot(valid && k >= pk);
            if (lane == 0) { M.mL[c] = mL; M.mR[c] = mR; }
        }
    }
    __syncthreads();
    plap(0);
    // S: per chunk the left stoppers before it and the right stoppers after it.  A thread owns up to 8 consecutive chunks and reads their masks
    // once (in the global form these are memory loads: all in flight together)
    {
        const int cpt = (nch + kSsThreads - 1) / kSsThreads, c0 = tid * cpt;
        int kl[8], kr[8], cl = 0, cr = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c = c0 + q;
            unsigned long long l = 0ull, r = 0ull;
            if (q < cpt && c < nch) { l = M.mL[c]; r = M.mR[c]; }
            kl[q] = __popcll(l); kr[q] = __popcll(r);
            cl += kl[q]; cr += kr[q];
        }
        const int il = wave_incl_scan(cl, lane), ir = wave_incl_scan(cr, lane);
        if (lane == 63) { ctl->wsum[0][wv] = il; ctl->wsum[1][wv] = ir; }
        __syncthreads();
        int bl = 0, br = 0, totR = 0;
        for (int w = 0; w < kSsWaves; ++w) { const int sl = ctl->wsum[0][w], sr = ctl->wsum[1][w]; if (w < wv) { bl += sl; br += sr; } totR += sr; }
        int runL = bl + il - cl, runR = br + ir - cr;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c = c0 + q;
            if (q < cpt && c < nch) {
                M.pA[c] = (uint32_t)runL;
                runL += kl[q]; runR += kr[q];
                M.pBx[c] = (uint32_t)(totR - runR);
                M.pBi[c] = (uint32_t)(totR - runR + kr[q]);      // right stoppers in this chunk and after it
            }
        }
        if (tid == kSsThreads - 1) ctl->bc[2] = runL;           // all left stoppers
    }
    __syncthreads();
    plap(1);
    if (wv == 0) {   // the crossing: m, the cut, the median's swap
        const int totL = ctl->bc[2];
        const int cs = wave_count_prefix(nch, lane, [&](int c) { return M.pA[c] <= M.pBi[c]; }) - 1;
        const unsigned long long sL = M.mL[cs], sR = M.mR[cs];
        const int sA = (int)M.pA[cs], sB = (int)M.pBi[cs];
        int v = min(sA + __popcll(sL & lanes_below(lane)), sB - __popcll(sR & lanes_below(lane)));
        const int m = max(wave_max(v), min(sA + __popcll(sL), sB - __popcll(sR)));
        int cut;
        if (m == 0) {
            const int c = wave_count_prefix(nch, lane, [&](int c) { return (c + 1 < nch ? (int)M.pA[c + 1] : totL) == 0; });   // chunks before the first left stopper
            cut = p0 + (c << 6) + __builtin_ctzll(M.mL[c]);
        } else {
            int Lm = 0x7fffffff;
            if (m < totL) {
                const int c = wave_count_prefix(nch, lane, [&](int c) { return (int)M.pA[c] <= m; }) - 1;
                Lm = p0 + (c << 6) + select64(M.mL[c], m - (int)M.pA[c]);
            }
            const int c = wave_count_prefix(nch, lane, [&](int c) { return (int)M.pBx[c] > m - 1; });
            cut = min(Lm, p0 + (c << 6) + select64_top(M.mR[c], m - 1 - (int)M.pBx[c]));
        }
        if (lane == 0) { a[first] = em; a[m3] = e0; ctl->bc[0] = m; ctl->bc[1] = cut; ctl->bc[3] = (int)pk; }
    }
    __syncthreads();
    plap(2);
    const int m = ctl->bc[0], cut = ctl->bc[1];
    // B: the right stoppers of rank < m leave their positions in the rank-indexed list (chunks from the right end; those further left only hold higher ranks)
    for (int c0 = nch - 1 - wv * U; c0 >= 0; c0 -= kSsWaves * U) {
        if ((int)M.pBx[c0] >= m) break;
        unsigned long long mR[U]; int bx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int c = c0 - u; mR[u] = c >= 0 ? M.mR[c] : 0ull; bx[u] = c >= 0 ? (int)M.pBx[c] : m; }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if ((mR[u] >> lane) & 1ull) {
                const int rank = bx[u] + __popcll(bits_above(mR[u], lane));
                if (rank < m) rpos[rank] = (rpos_t)(p0 + ((c0 - u) << 6) + lane);
            }
    }
    __syncthreads();
    plap(3);
    // C: the left stoppers of rank < m swap with their partners
    for (int c0 = wv * U; c0 < nch; c0 += kSsWaves * U) {
        if ((int)M.pA[c0] >= m) break;
        int i[U], j[U]; uint32_t vi[U], vj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u;
            i[u] = -1;
            if (c < nch) {
                const unsigned long long mL = M.mL[c];
                const int rank = (int)M.pA[c] + __popcll(mL & lanes_below(lane));
                if (((mL >> lane) & 1ull) && rank < m) i[u] = rank;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) if (i[u] >= 0) { j[u] = (int)rpos[i[u]]; i[u] = p0 + ((c0 + u) << 6) + lane; }
#pragma unroll
        for (int u = 0; u < U; ++u) if (i[u] >= 0) { vi[u] = a[i[u]]; vj[u] = a[j[u]]; }
#pragma unroll
        for (int u = 0; u < U; ++u) if (i[u] >= 0) { a[i[u]] = vj[u]; a[j[u]] = vi[u]; }
    }
    __syncthreads();
    plap(4);
    if (prof && threadIdx.x == 0) ctl->t[(G ? 16 : 10) + 5] += 1;
    return cut;
}

// One wave partitions [first, last) of the LDS window, 65 .. kSsTask entries: chunk c's masks and prefix sums live in lane c.
// Returns the cut, or -1 when the segment held equal keys and was FINISHED here (seed_sort_model.hpp uniform_final_pos): `depth` partitions are
// left in the recursion budget, the shortcut is taken only where the library would not have run out of it either.
__device__ __forceinline__ int wave_partition(uint32_t* a, lrpos_t* rpos_all, int first, int last, int depth, int lane, uint32_t skip_key, uint32_t* pivot_key) {
    lrpos_t* rpos = rpos_all + (first >> 1);
    const int p0 = first + 1, np = last - p0, nch = (np + 63) >> 6, mid = first + (last - first) / 2;
    const uint32_t e0 = a[first], ea = a[p0], eb = a[mid], ec = a[last - 1];
    const int m3 = median3_pos(ea >> kSsShift, eb >> kSsShift, ec >> kSsShift, p0, mid, last - 1);
    const uint32_t em = m3 == p0 ? ea : (m3 == mid ? eb : ec), pk = em >> kSsShift;
    *pivot_key = pk;
    unsigned long long myL = 0, myR = 0;
    constexpr int U = 4;                                         // chunks per trip: their loads are in flight together
    for (int c0 = 0; c0 < nch; c0 += U) {
        uint32_t e[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int pos = p0 + ((c0 + u) << 6) + lane; e[u] = pos < last ? a[pos] : 0u; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + u, pos = p0 + (c << 6) + lane;
            const bool valid = pos < last;
            const uint32_t k = (pos == m3 ? e0 : e[u]) >> kSsShift;
            const unsigned long long mL = __ballot(valid && k <= pk), mR = __ballot(valid && k >= pk);
            if (lane == c) { myL = mL; myR = mR; }
        }
    }
    const int cl = __popcll(myL), cr = __popcll(myR);
    const int il = wave_incl_scan(cl, lane), ir = wave_incl_scan(cr, lane);
    const int totL = __builtin_amdgcn_readlane(il, 63), totR = __builtin_amdgcn_readlane(ir, 63);
    if (totL == np && totR == np && last - first <= seedsort::kUniformMax && seedsort::uniform_levels(last - first) <= depth) {
        if (pk < skip_key) return -1;                            // equal keys below the skip key: no defined pixel here
        constexpr int K = seedsort::kUniformMax / 64;
        uint32_t v[K]; int dst[K];
#pragma unroll
        for (int j = 0; j < K; ++j) { const int x = first + (j << 6) + lane; v[j] = x < last ? a[x] : 0u; }
#pragma unroll
        for (int j = 0; j < K; ++j) { const int x = first + (j << 6) + lane; dst[j] = x < last ? seedsort::uniform_final_pos(x, first, last) : -1; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // every entry is read before the first one is overwritten
#pragma unroll
        for (int j = 0; j < K; ++j) if (dst[j] >= 0) a[dst[j]] = v[j];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        return -1;
    }
    const int A = il - cl, Bx = totR - ir;                      // left stoppers before chunk `lane`, right stoppers after it
    const int cs = __builtin_amdgcn_readfirstlane(__popcll(__ballot(lane < nch && A <= Bx + cr)) - 1);
    const unsigned long long sL = readlane64(myL, cs), sR = readlane64(myR, cs);
    const int sA = __builtin_amdgcn_readlane(A, cs), sB = __builtin_amdgcn_readlane(Bx, cs) + __popcll(sR);
    const int v = min(sA + __popcll(sL & lanes_below(lane)), sB - __popcll(sR & lanes_below(lane)));
    const int m = __builtin_amdgcn_readfirstlane(max(wave_max(v), min(sA + __popcll(sL), sB - __popcll(sR))));
    int cut;
    if (m == 0) {
        const int c = __builtin_amdgcn_readfirstlane(__builtin_ctzll(__ballot(lane < nch && cl > 0)));
        cut = p0 + (c << 6) + __builtin_ctzll(readlane64(myL, c));
    } else {
        int Lm = 0x7fffffff;
        if (m < totL) {
            const int c = __builtin_amdgcn_readfirstlane(__popcll(__ballot(lane < nch && A <= m)) - 1);
            Lm = p0 + (c << 6) + select64(readlane64(myL, c), m - __builtin_amdgcn_readlane(A, c));
        }
        const int c = __builtin_amdgcn_readfirstlane(__popcll(__ballot(lane < nch && Bx > m - 1)));
        cut = min(Lm, p0 + (c << 6) + select64_top(readlane64(myR, c), m - 1 - __builtin_amdgcn_readlane(Bx, c)));
    }
    if (lane == 0) { a[first] = em; a[m3] = e0; }
    for (int c = nch - 1; c >= 0; --c) {
        const int bx = __builtin_amdgcn_readlane(Bx, c);
        if (bx >= m) break;
        const unsigned long long mR = readlane64(myR, c);
        if ((mR >> lane) & 1ull) {
            const int rank = bx + __popcll(bits_above(mR, lane));
            if (rank < m) rpos[rank] = (lrpos_t)(p0 + (c << 6) + lane);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int c0 = 0; c0 < nch; c0 += U) {
        if (__builtin_amdgcn_readlane(A, c0) >= m) break;
        int i[U], j[U]; uint32_t vi[U], vj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = min(c0 + u, 63);
            const unsigned long long mL = c0 + u < nch ? readlane64(myL, c) : 0ull;
            const int rank = __builtin_amdgcn_readlane(A, c) + __popcll(mL & lanes_below(lane));
            i[u] = (((mL >> lane) & 1ull) && rank < m) ? rank : -1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) if (i[u] >= 0) { j[u] = (int)rpos[i[u]]; i[u] = p0 + ((c0 + u) << 6) + lane; }
#pragma unroll
        for (int u = 0; u < U; ++u) if (i[u] >= 0) { vi[u] = a[i[u]]; vj[u] = a[j[u]]; }
#pragma unroll
        for (int u = 0; u < U; ++u) if (i[u] >= 0) { a[i[u]] = vj[u]; a[j[u]] = vi[u]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    return cut;
}

// One lane: std::__introsort_loop on [first, last) of the LDS window, 17 .. kSsSmall entries, `depth` partitions left in the budget.
// The larger part waits, the smaller is continued: at most two parts wait at any time (each has more than 16 of at most 64 entries).
__device__ __forceinline__ void lane_introsort(uint32_t* v, int first, int last, int depth, uint32_t skip_key) {
    const KeyDesc comp;
    int pf0 = 0, pl0 = 0, pd0 = 0, pf1 = 0, pl1 = 0, pd1 = 0, np = 0;
    while (true) {
        while (last - first > 16) {
            if (depth == 0) { libstdcxx::heap_sort(v, first, last, comp); break; }
            --depth;
            const int mid = first + (last - first) / 2, a = first + 1, c = last - 1;
            const int m3 = median3_pos(v[a] >> kSsShift, v[mid] >> kSsShift, v[c] >> kSsShift, a, mid, c);
            { const unsigned t = v[first]; v[first] = v[m3]; v[m3] = t; }
            const unsigned pv = v[first];
            int lo = first + 1, hi = last;
            while (true) {
                while (comp(v[lo], pv)) ++lo;
                --hi;
                while (comp(pv, v[hi])) --hi;
                if (!(lo < hi)) break;
                const unsigned t = v[lo]; v[lo] = v[hi]; v[hi] = t;
                ++lo;
            }
            if ((pv >> kSsShift) < skip_key) { last = lo; continue; }   // the right part holds no defined pixel
            int wf, wl;                                          // the part that waits
            if (lo - first >= last - lo) { wf = first; wl = lo; first = lo; } else { wf = lo; wl = last; last = lo; }
            if (wl - wf > 16) {
                if (np == 0) { pf0 = wf; pl0 = wl; pd0 = depth; } else { pf1 = wf; pl1 = wl; pd1 = depth; }
                ++np;
            }
        }
        if (np == 0) break;
        --np;
        if (np == 0) { first = pf0; last = pl0; depth = pd0; } else { first = pf1; last = pl1; depth = pd1; }
    }
}

// ---- queues of the window phase (LDS; lane 0 of a wave, or thread 0, calls these) -----------------------------------------------
__device__ __forceinline__ int lds_add(int* p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int lds_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }

struct SsWin {
    uint32_t* win; lrpos_t* rpos; unsigned long long* q; uint32_t* small; SsCtl* ctl; int* dbg;
    uint32_t skip_key;               // a part whose keys are all below this holds no defined pixel: nothing in it is ever a seed, it is left alone
};
// a child of a partition inside the window: by size to the workgroup's stack, the wave tasks, the lanes' list, or nowhere
__device__ __forceinline__ void route(const SsWin& W, int first, int last, int depth) {
    const int n = last - first;
    if (n <= 16) return;
    if (n <= kSsSmall) {
        const int i = lds_add(&W.ctl->n_small, 1);
        if (i < kSsSmallCap) W.small[i] = (uint32_t)first | ((uint32_t)(n - 1) << 16) | ((uint32_t)depth << 24);
        else W.ctl->overflow = 1;
    } else if (n <= kSsTask) {
        lds_add(&W.ctl->q_open, 1);
        const int t = lds_add(&W.ctl->q_tail, 1);
        const unsigned long long rec = (unsigned long long)(uint32_t)first | ((unsigned long long)(uint32_t)last << 16) | ((unsigned long long)(uint32_t)depth << 32) | (1ull << 63);
        __hip_atomic_store(&W.q[t & (kSsQueue - 1)], rec, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        lds_add(&W.ctl->q_pending, 1);
    } else {
        const int i = W.ctl->w_n;                                // thread 0 only: whole-workgroup phase
        if (i < kSsWStack) { W.ctl->w_first[i] = first; W.ctl->w_last[i] = last; W.ctl->w_depth[i] = depth; W.ctl->w_n = i + 1; }
        else W.ctl->overflow = 1;
    }
}

// The segment ent[gfirst, glast) (17 .. kSsT entries) is finished in LDS.  All threads; starts and ends with barriers.
__device__ __forceinline__ void window_process(uint32_t* ent, int gfirst, int glast, int depth, const SsWin& W, const SsMem& ML) {
    const int tid = threadIdx.x, lane = tid & 63, n = glast - gfirst;
    SsCtl* ctl = W.ctl;
    long long t0 = W.dbg ? clock64() : 0;
    auto lap = [&](int k) { if (W.dbg && tid == 0) { const long long t1 = clock64(); ctl->t[k] += t1 - t0; t0 = t1; } };
    for (int i = tid; i < n; i += kSsThreads) W.win[i] = ent[gfirst + i];
    if (tid == 0) { ctl->w_n = 0; ctl->q_head = ctl->q_tail = ctl->q_pending = ctl->q_open = 0; ctl->n_small = 0; }
    for (int i = tid; i < kSsQueue; i += kSsThreads) W.q[i] = 0ull;
    __syncthreads();
    lap(1);
    if (tid == 0) route(W, 0, n, depth);
    while (true) {   // whole-workgroup partitions
        __syncthreads();
        const int sn = ctl->w_n;
        if (sn == 0) break;
        const int first = ctl->w_first[sn - 1], last = ctl->w_last[sn - 1], d = ctl->w_depth[sn - 1];
        __syncthreads();
        if (d == 0) {
            if (tid == 0) { libstdcxx::heap_sort(W.win, first, last, KeyDesc()); ctl->w_n = sn - 1; }
            continue;
        }
        const int cut = wg_partition<false>(ML, ctl, first, last, W.dbg != nullptr);
        if (tid == 0) { ctl->w_n = sn - 1; route(W, first, cut, d - 1); if ((uint32_t)ctl->bc[3] >= W.skip_key) route(W, cut, last, d - 1); }   // right part: keys <= pivot
    }
    lap(2);
    // wave tasks.  The loop is UNIFORM: every lane iterates; lane 0 alone touches the queue counters, in straight-line code whose results
    // are broadcast, and every lane reads the claimed slot itself (same address).  (An earlier version let lane 0 spin in a loop of its own
    // and handed the record over from variables only lane 0 assigned: the compiler let the other lanes leave the task loop on their own
    // and every wave's SECOND partition ran with one lane -- profiles/r04_seed_sort.md.)
    const int wv = tid >> 6;
    for (int spins = 0;; ++spins) {
        int st = 0, slot = 0;                                    // 0: nothing to take right now, 1: claimed `slot`, 2: every task is finished
        if (lane == 0) {
            if (lds_load(&ctl->q_open) == 0) st = 2;
            else if (lds_add(&ctl->q_pending, -1) > 0) { slot = lds_add(&ctl->q_head, 1) & (kSsQueue - 1); st = 1; }
            else lds_add(&ctl->q_pending, 1);
        }
        st = __builtin_amdgcn_readfirstlane(st); slot = __builtin_amdgcn_readfirstlane(slot);
        if (st == 2) break;
        if (spins > (1 << 22)) { if (lane == 0) ctl->overflow = 1; break; }   // (a protocol error becomes a status instead of a hang)
        if (st == 0) { __builtin_amdgcn_s_sleep(4); continue; }
        unsigned rlo = 0, rhi = 0;
        for (int w = 0; w < (1 << 22); ++w) {                    // the pusher holding this ticket writes the slot before it counts the task as pending
            const unsigned long long rec = __hip_atomic_load(&W.q[slot], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            rlo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)rec); rhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(rec >> 32));
            if (rhi >> 31) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (!(rhi >> 31)) { if (lane == 0) ctl->overflow = 1; break; }
        if (lane == 0) __hip_atomic_store(&W.q[slot], 0ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int first = (int)(rlo & 0xffffu), last = (int)(rlo >> 16), d = (int)(rhi & 0xffu);
        const long long tb = W.dbg ? clock64() : 0;
        if (d == 0) {
            if (lane == 0) libstdcxx::heap_sort(W.win, first, last, KeyDesc());
        } else {
            uint32_t pk = 0;
            const int cut = wave_partition(W.win, W.rpos, first, last, d, lane, W.skip_key, &pk);
            if (W.dbg && lane == 0) { const int k = atomicAdd(&W.dbg[1], 1); if (k < 4000) { int* r = W.dbg + 2 + 6 * k; r[0] = wv; r[1] = first; r[2] = last; r[3] = d; r[4] = cut; r[5] = (int)clock(); } }
            if (lane == 0 && cut >= 0) { route(W, first, cut, d - 1); if (pk >= W.skip_key) route(W, cut, last, d - 1); }
        }
        if (W.dbg && lane == 0) { const long long dt = clock64() - tb; atomicAdd((unsigned long long*)&ctl->t[8], (unsigned long long)dt); atomicMax((unsigned long long*)&ctl->t[9], (unsigned long long)dt); }
        if (lane == 0) lds_add(&ctl->q_open, -1);
        spins = 0;
    }
    __syncthreads();
    lap(3);
    const int ns = min(ctl->n_small, kSsSmallCap);
    for (int i = tid; i < ns; i += kSsThreads) {
        const uint32_t r = W.small[i];
        const int first = (int)(r & 0xffffu);
        lane_introsort(W.win, first, first + (int)((r >> 16) & 0xffu) + 1, (int)(r >> 24), W.skip_key);
    }
    __syncthreads();
    lap(4);
    for (int i = tid; i < n; i += kSsThreads) ent[gfirst + i] = W.win[i];
    __syncthreads();
    lap(5);
    if (W.dbg && tid == 0) ctl->t[6] += 1;
}

// std::__introsort_loop(ent, ent + n, depth) by the workgroup.  ws: the frame's scratch (partner positions, chunk masks).
__device__ __forceinline__ void seed_introsort_loop(uint32_t* ent, int n, int depth, uint32_t skip_key, uint32_t* ws, unsigned char* lds, int32_t* status, int* dbg = nullptr) {
    const int tid = threadIdx.x;
    SsCtl* ctl = (SsCtl*)(lds + kOffCtl);
    SsWin W{(uint32_t*)(lds + kOffWin), (lrpos_t*)(lds + kOffRpos), (unsigned long long*)(lds + kOffQ), (uint32_t*)(lds + kOffSmall), ctl, dbg, skip_key};
    const SsMem ML{W.win, W.rpos, (unsigned long long*)(lds + kOffML), (unsigned long long*)(lds + kOffMR), (uint32_t*)(lds + kOffPA), (uint32_t*)(lds + kOffPBx), (uint32_t*)(lds + kOffPBi)};
    const int nchg = (n + 63) / 64 + 1;
    uint32_t* g_rpos = ws;                                                                   // n / 2 + 64 entries
    unsigned long long* g_mL = (unsigned long long*)(ws + (((size_t)n / 2 + 64 + 1) & ~(size_t)1));
    const SsMem MG{ent, g_rpos, g_mL, g_mL + nchg, (uint32_t*)(lds + kOffWin), (uint32_t*)(lds + kOffWin) + kSsMaxChunksG, (uint32_t*)(lds + kOffWin) + 2 * kSsMaxChunksG};
    if (tid == 0) { ctl->g_n = 1; ctl->g_first[0] = 0; ctl->g_last[0] = n; ctl->g_depth[0] = depth; ctl->overflow = 0; for (int k = 0; k < 24; ++k) ctl->t[k] = 0; }
    while (true) {
        __syncthreads();
        const int sn = ctl->g_n;
        if (sn == 0) break;
        const int first = ctl->g_first[sn - 1], last = ctl->g_last[sn - 1], d = ctl->g_depth[sn - 1];
        __syncthreads();
        if (tid == 0) ctl->g_n = sn - 1;
        if (last - first <= 16) continue;
        if (last - first <= kSsT) { window_process(ent, first, last, d, W, ML); continue; }
        if (d == 0) {
            if (tid == 0) libstdcxx::heap_sort(ent, first, last, KeyDesc());
            continue;
        }
        const long long tg = dbg ? clock64() : 0;
        const int cut = wg_partition<true>(MG, ctl, first, last, dbg != nullptr);
        if (dbg && tid == 0) { ctl->t[0] += clock64() - tg; ctl->t[7] += 1; }
        if (tid == 0) {
            int i = ctl->g_n;
            if (i + 2 <= kSsGStack) {
                ctl->g_first[i] = first; ctl->g_last[i] = cut; ctl->g_depth[i] = d - 1; ++i;
                if ((uint32_t)ctl->bc[3] >= skip_key) { ctl->g_first[i] = cut; ctl->g_last[i] = last; ctl->g_depth[i] = d - 1; ++i; }
                ctl->g_n = i;
            } else ctl->overflow = 1;
        }
    }
    if (tid == 0 && ctl->overflow) atomicOr(status, 32);
    if (dbg && tid == 0) for (int k = 0; k < 24; ++k) { dbg[2 + 6 * 4000 + 2 * k] = (int)(ctl->t[k] & 0xffffffff); dbg[2 + 6 * 4000 + 2 * k + 1] = (int)(ctl->t[k] >> 32); }
}

}  // namespace

size_t seed_sort_ws_entries(size_t nv) { return ((nv / 2 + 64 + 1) & ~(size_t)1) + 2 * 2 * ((nv + 63) / 64 + 1); }   // u32 units per frame
size_t seed_sort_lds_bytes() { return kSsLdsBytes; }

// The frame's seed array: every pixel (x < sw - 1, y < sh - 1) row-major with its gradient bin (lsd.cpp ll_angle), then the
// introsort loop.  P.g2 holds gx^2 + gy^2 of EVERY such pixel in this mode (LsdParams::seed_exact).
__global__ __launch_bounds__(kSsThreads) void k_lsd_seed_sort(LinePlanes P, LsdParams lp, int n_grad_blocks, uint32_t* ent_all, uint32_t* ws_all, size_t ws_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ss_lds[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = P.sw * P.sh, nv = (P.sw - 1) * (P.sh - 1);
    SsCtl* ctl = (SsCtl*)(ss_lds + kOffCtl);
    uint32_t mx = 0;
    for (int i = tid; i < n_grad_blocks; i += kSsThreads) mx = max(mx, P.blockmax[(size_t)b * n_grad_blocks + i]);
    mx = (uint32_t)wave_max((int)mx);
    if (lane == 0) ctl->wsum[0][wv] = (int)mx;
    __syncthreads();
    mx = 0;
    for (int w = 0; w < kSsWaves; ++w) mx = max(mx, (uint32_t)ctl->wsum[0][w]);
    __syncthreads();
    const double max_grad = mx ? sqrt((double)mx / 4.0) : -1.0;
    const double bin_coef = (max_grad > 0) ? (double)(lp.n_bins - 1) / max_grad : 0;
    const uint32_t* g2 = P.g2 + (size_t)b * n;
    uint32_t* ent = ent_all + (size_t)b * nv;
    for (int y = wv; y < P.sh - 1; y += kSsWaves)
        for (int x = lane; x < P.sw - 1; x += 64) {
            const uint32_t pix = (uint32_t)(y * P.sw + x), v = g2[pix];
            const uint32_t bin = (uint32_t)(int)(sqrt((double)v / 4.0) * bin_coef);
            ent[y * (P.sw - 1) + x] = pix | (v >= lp.g2_def_min ? kSsDefBit : 0u) | (bin << kSsShift);
        }
    int lg = 0;
    while ((2 << lg) <= nv) ++lg;
    // Undefined pixels are sorted with the rest (they take part in every partition) but are never seeds: a part that can hold only them
    // -- every key below the bin of the smallest defined magnitude -- is left as it is (70 % of a frame)
    const uint32_t skip_key = (uint32_t)(int)(sqrt((double)lp.g2_def_min / 4.0) * bin_coef);
    seed_introsort_loop(ent, nv, 2 * lg, skip_key, ws_all + (size_t)b * ws_stride, ss_lds, P.status);
}

// test entry: the introsort loop on caller-made entries with a chosen recursion budget
__global__ __launch_bounds__(kSsThreads) void k_seed_sort_debug(uint32_t* ent, int n, int depth, uint32_t skip_key, uint32_t* ws, int32_t* status, int* dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ss_lds[];
    if (n > 16) seed_introsort_loop(ent, n, depth, skip_key, ws, ss_lds, status, dbg);
}

// The final insertion sort = a stable counting sort by bin of the DEFINED entries, in array order (k_lsd_order's steps 2-4 on the
// permuted array instead of the row-major pixel sequence).
__global__ __launch_bounds__(256) void k_lsd_order_entries(LinePlanes P, const uint32_t* __restrict__ ent_all) {
    __shared__ uint32_t cnt[4][1024];
    __shared__ uint32_t s_wsum[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
    const int nv = (P.sw - 1) * (P.sh - 1);
    for (int i = tid; i < 4096; i += 256) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const uint32_t* ent = ent_all + (size_t)b * nv;
    uint32_t* order = P.order + (size_t)b * nv;
    const int ngroups = (nv + 63) / 64, gper = (ngroups + 3) / 4, g0 = q * gper, g1 = min(ngroups, g0 + gper);
    uint32_t* comp = P.reg + (size_t)b * P.reg_frame_stride + (size_t)g0 * 64;
    int ncomp = 0;
    for (int gb = g0; gb < g1; gb += 4) {
        uint32_t v4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = (gb + u) * 64 + lane;
            v4[u] = (gb + u < g1 && i < nv) ? ent[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t v = v4[u];
            const unsigned long long defm = __ballot((v & kSsDefBit) != 0);
            if (!defm) continue;
            if (v & kSsDefBit) {
                atomicAdd(&cnt[q][v >> kSsShift], 1u);
                comp[ncomp + __popcll(defm & ((1ull << lane) - 1ull))] = v & ~kSsDefBit;
            }
            ncomp += __popcll(defm);
        }
    }
    __syncthreads();
    {   // thread t owns bins 1023-4t .. 1020-4t (descending)
        const int v0 = 1023 - 4 * tid;
        uint32_t c[4][4], tot = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) { c[j][k] = cnt[k][v0 - j]; tot += c[j][k]; }
        uint32_t inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, o); if (lane >= o) inc += t; }
        if (lane == 63) s_wsum[q] = inc;
        __syncthreads();
        uint32_t run = inc - tot;
        for (int k = 0; k < q; ++k) run += s_wsum[k];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) { cnt[k][v0 - j] = run; run += c[j][k]; }
        if (tid == 255) P.n_order[b] = (int32_t)run;
    }
    __syncthreads();
    for (int ib = 0; ib < ncomp; ib += 256) {
        uint32_t e4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) e4[u] = ib + 64 * u + lane < ncomp ? comp[ib + 64 * u + lane] : 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i0 = ib + 64 * u;
            if (i0 >= ncomp) break;
            const bool valid = i0 + lane < ncomp;
            const uint32_t e = e4[u];
            const unsigned v = e >> kSsShift;
            unsigned long long peers = __ballot(valid);
#pragma unroll
            for (int bit = 0; bit < 10; ++bit) {
                const unsigned long long m = __ballot((v >> bit) & 1u);
                peers &= ((v >> bit) & 1u) ? m : ~m;
            }
            if (valid) {
                const int rank = __popcll(peers & ((1ull << lane) - 1ull));
                order[cnt[q][v] + rank] = e & kLsdSeedPixMask;
            }
            __builtin_amdgcn_wave_barrier();
            if (valid && (peers & ((1ull << lane) - 1ull)) == 0) cnt[q][v] += (uint32_t)__popcll(peers);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

void launch_seed_order_exact(hipStream_t st, const LinePlanes& P, const LsdParams& lp, int B, uint32_t* ent, uint32_t* ws, size_t ws_stride) {
    const int n = P.sw * P.sh;
    hipLaunchKernelGGL(k_lsd_seed_sort, dim3(B), dim3(kSsThreads), kSsLdsBytes, st, P, lp, (n + 255) / 256, ent, ws, ws_stride);
    hipLaunchKernelGGL(k_lsd_order_entries, dim3(B), dim3(256), 0, st, P, (const uint32_t*)ent);
}
hipError_t seed_sort_configure() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_lsd_seed_sort), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSsLdsBytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_seed_sort_debug), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSsLdsBytes);
}
void launch_seed_sort_debug(hipStream_t st, uint32_t* ent, int n, int depth, uint32_t skip_key, uint32_t* ws, int32_t* status, int* dbg) {
    hipLaunchKernelGGL(k_seed_sort_debug, dim3(1), dim3(kSsThreads), kSsLdsBytes, st, ent, n, depth, skip_key, ws, status, dbg);
}

}  // namespace plp
