// The permutation libstdc++'s std::sort leaves on LSD's seed array, as a DATA-PARALLEL formulation (closes definition D1).
//
// OpenCV's LSD (lsd.cpp ll_angle, reached from the reference's LSDDetector_custom.cpp:244-257) sorts every pixel of the scaled
// image by gradient bin with std::sort and a comparator that looks at the bin only: the order inside a bin is whatever the
// library's introsort leaves.  A reference built with GCC therefore visits seeds in the order of
//     std::__introsort_loop (median-of-three pivot, unguarded Hoare partition, recursion budget 2 * floor(log2 n), heap sort when
//     it runs out)  +  std::__final_insertion_sort
// and region growing depends on that order.  Two facts make it computable in parallel:
//
//  (1) The final insertion sort is a STABLE sort of whatever the introsort loop leaves (it only moves an element left past strictly
//      "later" ones), so: result = stable sort by bin of the post-introsort array.  The kernels already have that counting sort.
//
//  (2) One unguarded Hoare partition of [first + 1, last) around the pivot v[first] is a PAIRING BY RANK.  The left scan stops at
//      elements that do not go before the pivot (key <= pivot for the descending comparator), the right scan at elements the pivot
//      does not go before (key >= pivot).  Neither scan ever re-examines a swapped element, so the k-th swap exchanges
//          L[k] = the k-th left stopper from the left   with   R[k] = the k-th right stopper from the right   (original array),
//      for k < m = #{k : L[k] < R[k]} (L increases, R decreases: the condition is monotone in k), and the returned cut is
//          m == 0 ? L[0] : min(L[m], R[m - 1])      (L[m] = +inf if there is no such stopper; after m swaps position R[m - 1]
//                                                     holds a left stopper).
//      With C(x) = (#left stoppers before x, #right stoppers at or after x), m = max over x of min of the two: one crossing.
//      Stoppers are found by ballots, ranks by prefix sums over 64-element chunks, partners through a rank-indexed position list.
//
// This header is the host model of exactly that arithmetic (chunk masks, chunk prefix sums, the crossing chunk, bit selection) in
// plain C++: the CPU suite checks it against the real std::sort and against libstdc++'s own __introsort_loop with forced recursion
// budgets (tests/test_index_models.py), and the GPU tests check the kernels (seed_sort_kernels.hip) against it and the oracle.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "libstdcxx_sort.hpp"

namespace plp {
namespace seedsort {

constexpr int kKeyShift = 20;                 // entry = pixel | defined << 19 | bin << 20 (line_device.hpp kLsdSeedPixBits)
struct KeyDesc { PLP_SORT_HD bool operator()(unsigned a, unsigned b) const { return (a >> kKeyShift) > (b >> kKeyShift); } };

// std::__move_median_to_first's choice among positions a, b, c for the comparator "larger key first" (ka, kb, kc: their keys)
PLP_SORT_HD int median3_pos(uint32_t ka, uint32_t kb, uint32_t kc, int a, int b, int c) {
    if (ka > kb) return (kb > kc) ? b : ((ka > kc) ? c : a);
    return (ka > kc) ? a : ((kb > kc) ? c : b);
}
// position of the k-th (0-based) set bit of m, counted from bit 0; m has more than k bits set
PLP_SORT_HD int select64(unsigned long long m, int k) {
    int pos = 0;
    for (int w = 32; w >= 1; w >>= 1) {
        const unsigned long long low = m & ((1ull << w) - 1ull);
#if defined(__HIP_DEVICE_COMPILE__)
        const int c = __popcll(low);
#else
        const int c = __builtin_popcountll(low);
#endif
        if (k >= c) { k -= c; m >>= w; pos += w; }
    }
    return pos;
}
PLP_SORT_HD unsigned long long bitrev64(unsigned long long v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brevll(v);
#else
    unsigned long long r = 0;
    for (int i = 0; i < 64; ++i) r |= ((v >> i) & 1ull) << (63 - i);
    return r;
#endif
}
// A segment [f, l) whose keys are ALL EQUAL needs no comparisons at all: the median of three is the middle candidate, every element stops both
// scans, the k-th swap exchanges p0 + k with l - 1 - k for k < m = (l - f - 1) / 2, the cut is p0 + m -- and both parts are segments of
// equal keys again.  So the place where the element at x ends after the whole std::__introsort_loop of the segment is a function of (x, f, l)
// alone (87 % of a frame's seed array ends in ~500 such segments; 79 % of all partitions of the replay are partitions of equal keys).
PLP_SORT_HD int uniform_final_pos(int x, int f, int l) {
    while (l - f > 16) {
        const int p0 = f + 1, np = l - p0, mid = f + (l - f) / 2, m = np >> 1, cut = p0 + m;
        if (x == f) x = mid; else if (x == mid) x = f;                                        // the median's swap
        if (x >= p0 && (x - p0 < m || l - 1 - x < m)) x = p0 + l - 1 - x;                     // the k-th pair
        if (x < cut) l = cut; else f = cut;
    }
    return x;
}
// partitions on the longest path below a segment of n equal keys (the left part, 1 + (n - 1) / 2 entries, is never the shorter one)
PLP_SORT_HD int uniform_levels(int n) {
    int levels = 0;
    while (n > 16) { n = 1 + (n - 1) / 2; ++levels; }
    return levels;
}
constexpr int kUniformMax = 512;   // the kernels finish a segment of equal keys in one step up to this length (8 entries per lane)

// position of the k-th set bit counted from bit 63 downwards
PLP_SORT_HD int select64_top(unsigned long long m, int k) { return 63 - select64(bitrev64(m), k); }

// ---- host model ------------------------------------------------------------------------------------------------------------
inline int popc64(unsigned long long v) { return __builtin_popcountll(v); }

// One partition of [first, last) (more than 16 entries) in the chunked rank-pairing form the kernels use; returns the cut.
inline int partition_model(uint32_t* v, int first, int last) {
    const int p0 = first + 1, np = last - p0, nch = (np + 63) >> 6, mid = first + (last - first) / 2;
    const uint32_t e0 = v[first];
    const int m3 = median3_pos(v[p0] >> kKeyShift, v[mid] >> kKeyShift, v[last - 1] >> kKeyShift, p0, mid, last - 1);
    const uint32_t pk = v[m3] >> kKeyShift;
    // step A: chunk masks of the array as it is AFTER the median's swap with v[first] (done physically below)
    std::vector<unsigned long long> mL(nch, 0), mR(nch, 0);
    for (int c = 0; c < nch; ++c)
        for (int l = 0; l < 64; ++l) {
            const int pos = p0 + c * 64 + l;
            if (pos >= last) break;
            const uint32_t k = (pos == m3 ? e0 : v[pos]) >> kKeyShift;
            if (k <= pk) mL[c] |= 1ull << l;
            if (k >= pk) mR[c] |= 1ull << l;
        }
    // step S: chunk prefix sums; the crossing chunk; m; the cut
    std::vector<int> pA(nch), pBx(nch);
    int totL = 0, totR = 0;
    for (int c = 0; c < nch; ++c) { pA[c] = totL; totL += popc64(mL[c]); totR += popc64(mR[c]); }
    for (int c = 0, run = 0; c < nch; ++c) { run += popc64(mR[c]); pBx[c] = totR - run; }
    int n_true = 0;
    while (n_true < nch && pA[n_true] <= pBx[n_true] + popc64(mR[n_true])) ++n_true;
    const int cs = n_true - 1;                                   // A_c <= B_c holds at chunk 0 (0 <= totR)
    int m = 0;
    for (int t = 0; t <= 64; ++t) {
        const unsigned long long low = t == 64 ? ~0ull : (1ull << t) - 1ull;
        m = std::max(m, std::min(pA[cs] + popc64(mL[cs] & low), pBx[cs] + popc64(mR[cs]) - popc64(mR[cs] & low)));
    }
    int cut;
    if (m == 0) {
        int c = 0;
        while (!mL[c]) ++c;
        cut = p0 + c * 64 + __builtin_ctzll(mL[c]);
    } else {
        int Lm = 0x7fffffff;
        if (m < totL) {
            int c = 0;
            while (c + 1 < nch && pA[c + 1] <= m) ++c;
            Lm = p0 + c * 64 + select64(mL[c], m - pA[c]);
        }
        int c = 0;
        while (pBx[c] > m - 1) ++c;                               // first chunk whose later chunks hold at most m - 1 right stoppers
        const int Rm = p0 + c * 64 + select64_top(mR[c], m - 1 - pBx[c]);
        cut = std::min(Lm, Rm);
    }
    // the median's swap, then steps B and C: partner positions by rank, swaps
    v[first] = v[m3]; v[m3] = e0;
    std::vector<int> rpos(m);
    for (int c = 0; c < nch; ++c)
        for (int l = 0; l < 64; ++l)
            if ((mR[c] >> l) & 1ull) {
                const int rank = pBx[c] + popc64(l == 63 ? 0ull : mR[c] >> (l + 1));
                if (rank < m) rpos[rank] = p0 + c * 64 + l;
            }
    for (int c = 0; c < nch; ++c)
        for (int l = 0; l < 64; ++l)
            if ((mL[c] >> l) & 1ull) {
                const int rank = pA[c] + popc64(mL[c] & ((1ull << l) - 1ull));
                if (rank < m) std::swap(v[p0 + c * 64 + l], v[rpos[rank]]);
            }
    return cut;
}

// std::__introsort_loop(v, v + n, depth_limit) in that form.  depth_limit < 0: the library's 2 * floor(log2 n).
// skip_key: the kernels leave alone every part that can only hold keys below it (the right part of a partition whose pivot key is below
// it): in the seed array those are undefined pixels, which are sorted along but never become seeds; 0 = sort everything.
inline void introsort_loop_model(uint32_t* v, int n, int depth_limit = -1, uint32_t skip_key = 0) {
    if (n <= 16) return;
    int lg = 0;
    while ((2 << lg) <= n) ++lg;
    struct Seg { int first, last, depth; };
    std::vector<Seg> st{{0, n, depth_limit < 0 ? 2 * lg : depth_limit}};
    while (!st.empty()) {
        const Seg s = st.back();
        st.pop_back();
        if (s.last - s.first <= 16) continue;
        if (s.depth == 0) { libstdcxx::heap_sort(v, s.first, s.last, KeyDesc()); continue; }
        const int n_seg = s.last - s.first;
        if (n_seg <= kUniformMax && uniform_levels(n_seg) <= s.depth) {   // the kernels' shortcut for a segment of equal keys
            bool uniform = true;
            for (int i = s.first + 1; i < s.last && uniform; ++i) uniform = (v[i] >> kKeyShift) == (v[s.first] >> kKeyShift);
            if (uniform) {
                if ((v[s.first] >> kKeyShift) < skip_key) continue;                                  // equal keys below the skip key: left alone
                std::vector<uint32_t> src(v + s.first, v + s.last);
                for (int x = s.first; x < s.last; ++x) v[uniform_final_pos(x, s.first, s.last)] = src[x - s.first];
                continue;
            }
        }
        const int cut = partition_model(v, s.first, s.last);
        st.push_back({s.first, cut, s.depth - 1});
        if ((v[s.first] >> kKeyShift) >= skip_key) st.push_back({cut, s.last, s.depth - 1});      // v[first] is the pivot now
    }
}

}  // namespace seedsort
}  // namespace plp
