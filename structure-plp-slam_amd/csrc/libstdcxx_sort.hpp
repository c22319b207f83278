// std::sort as libstdc++ implements it (bits/stl_algo.h: introsort with median-of-three pivots down to chunks of 16, heap sort
// when the recursion budget 2 * floor(log2 n) runs out, then one insertion sort), restated for arrays of at most 64 elements.
//
// Why: match::angle_checker (reference src/PLPSLAM/match/angle_checker.h:165-176) ranks its 30 histogram bins with std::sort and a
// comparator that looks at the bin SIZE only, and keeps the matches of the first three.  Among equally full bins the order is
// whatever the library's algorithm produces -- deterministic, but not stable -- and it decides which matches survive whenever the
// third- and fourth-fullest bins tie.  Reproducing the algorithm (rather than defining a tie rule of our own) makes the result
// equal to a reference built with GCC's library, ties included.  The CPU suite checks this restatement against the real std::sort
// on a million random and adversarial inputs (tests/test_abi_and_model.py).
// Plain C++ (no library calls): used by the matcher kernels (one thread) and by a host entry point for that test.
#pragma once

#if defined(__HIPCC__)
#define PLP_SORT_HD __host__ __device__ inline
#else
#define PLP_SORT_HD inline
#endif

namespace plp {
namespace libstdcxx {

// comp(a, b): "a goes before b".  Less::operator()(unsigned a, unsigned b) const.
template <class Less> PLP_SORT_HD void unguarded_linear_insert(unsigned* v, int last, const Less& comp) {
    const unsigned val = v[last];
    int next = last - 1;
    while (comp(val, v[next])) { v[last] = v[next]; last = next; --next; }
    v[last] = val;
}
template <class Less> PLP_SORT_HD void insertion_sort(unsigned* v, int first, int last, const Less& comp) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (comp(v[i], v[first])) {
            const unsigned val = v[i];
            for (int k = i; k > first; --k) v[k] = v[k - 1];   // move_backward(first, i, i + 1)
            v[first] = val;
        } else unguarded_linear_insert(v, i, comp);
    }
}
template <class Less> PLP_SORT_HD void push_heap_(unsigned* v, int first, int hole, int top, unsigned value, const Less& comp) {
    int parent = (hole - 1) / 2;
    while (hole > top && comp(v[first + parent], value)) { v[first + hole] = v[first + parent]; hole = parent; parent = (hole - 1) / 2; }
    v[first + hole] = value;
}
template <class Less> PLP_SORT_HD void adjust_heap(unsigned* v, int first, int hole, int len, unsigned value, const Less& comp) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (comp(v[first + child], v[first + child - 1])) --child;
        v[first + hole] = v[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        v[first + hole] = v[first + child - 1];
        hole = child - 1;
    }
    push_heap_(v, first, hole, top, value, comp);
}
template <class Less> PLP_SORT_HD void heap_sort(unsigned* v, int first, int last, const Less& comp) {   // __partial_sort(first, last, last)
    const int len = last - first;
    if (len >= 2)
        for (int parent = (len - 2) / 2;; --parent) {   // __make_heap
            adjust_heap(v, first, parent, len, v[first + parent], comp);
            if (parent == 0) break;
        }
    for (int l = last; l - first > 1;) {                // __sort_heap / __pop_heap
        --l;
        const unsigned value = v[l];
        v[l] = v[first];
        adjust_heap(v, first, 0, l - first, value, comp);
    }
}
// depth_limit < 0: the library's 2 * floor(log2 n); tests pass 0..3 to reach the heap-sort branch, which real inputs of 30 bins almost never do
// ws: 48 ints of scratch for the explicit stack (the kernels pass LDS: a thread-private array indexed at run time would put the
// whole kernel into scratch memory)
template <class Less> PLP_SORT_HD void sort(unsigned* v, int n, const Less& comp, int* ws, int depth_limit = -1) {
    if (n <= 0) return;
    int lg = 0;
    while ((2 << lg) <= n) ++lg;                        // std::__lg(n)
    // __introsort_loop; the recursion on [cut, last) becomes an explicit stack (the parts are disjoint: their order is irrelevant)
    int *st_first = ws, *st_last = ws + 16, *st_depth = ws + 32, sp = 0;
    st_first[0] = 0; st_last[0] = n; st_depth[0] = depth_limit < 0 ? 2 * lg : depth_limit; sp = 1;
    while (sp) {
        --sp;
        const int first = st_first[sp];
        int last = st_last[sp], depth = st_depth[sp];
        while (last - first > 16) {
            if (depth == 0) { heap_sort(v, first, last, comp); break; }
            --depth;
            // __unguarded_partition_pivot: median of (first + 1, mid, last - 1) moved to first, then the unguarded partition of [first + 1, last)
            const int mid = first + (last - first) / 2, a = first + 1, b = mid, c = last - 1;
            int m;
            if (comp(v[a], v[b])) m = comp(v[b], v[c]) ? b : (comp(v[a], v[c]) ? c : a);
            else m = comp(v[a], v[c]) ? a : (comp(v[b], v[c]) ? c : b);
            { const unsigned t = v[first]; v[first] = v[m]; v[m] = t; }
            int lo = first + 1, hi = last;
            while (true) {
                while (comp(v[lo], v[first])) ++lo;
                --hi;
                while (comp(v[first], v[hi])) --hi;
                if (!(lo < hi)) break;
                const unsigned t = v[lo]; v[lo] = v[hi]; v[hi] = t;
                ++lo;
            }
            st_first[sp] = lo; st_last[sp] = last; st_depth[sp] = depth; ++sp;   // __introsort_loop(cut, last, depth)
            last = lo;
        }
    }
    // __final_insertion_sort
    if (n > 16) {
        insertion_sort(v, 0, 16, comp);
        for (int i = 16; i != n; ++i) unguarded_linear_insert(v, i, comp);
    } else insertion_sort(v, 0, n, comp);
}

// the reference's index_sort_by_size: indices 0..n-1 ordered by std::sort with "bin a holds more than bin b"
struct BySizeDesc { const int* size; PLP_SORT_HD bool operator()(unsigned a, unsigned b) const { return size[a] > size[b]; } };
PLP_SORT_HD void index_sort_by_size(const int* size, int n, unsigned* idx, int* ws, int depth_limit = -1) {
    for (int i = 0; i < n; ++i) idx[i] = (unsigned)i;
    sort(idx, n, BySizeDesc{size}, ws, depth_limit);
}

}  // namespace libstdcxx
}  // namespace plp
