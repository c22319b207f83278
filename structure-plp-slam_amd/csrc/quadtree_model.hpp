// Host model of the device quadtree kernel (k_quadtree in orb_kernels.hip).
//
// The reference distributes key points with a std::list of nodes that are split level by
// level (src/PLPSLAM/feature/orb_extractor.cc:468-685, orb_extractor_node.cc:31-80).  The
// split positions depend only on the node rectangle (ceil(size/2)), never on the points,
// so every candidate's path through the tree is a pure function of its (x, y): a key of
// 2-bit child indices, most significant = first split, prefixed by the initial-node index.
// After ONE stable sort by that key every tree node is a contiguous range of the sorted
// array and "dividing a node" is three binary searches; no candidate is ever moved again.
// The list bookkeeping (push_front order, leaf flags of the initial nodes, the forkable-leaf
// pool sorted by (count, creation) in the fill phase) is replayed on node records only.
//
// This file is the sequential statement of exactly the steps the kernel runs in parallel;
// tests/test_quadtree_model.py checks it against the oracle's std::list restatement.
#pragma once
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

#include "orb_tables.hpp"

namespace plp {

constexpr int kQtDepth = 13;   // 2 bits per split; resolves rectangles up to 8192 px

struct QtCand { int x, y, score; };   // border-relative integer position + FAST score

inline uint32_t qt_key(int x, int y, const LevelGeom& L) {
    const unsigned ix = (unsigned)((float)x / L.delta_x);
    const unsigned iy = (unsigned)((float)y / L.delta_y);
    const unsigned node = ix + iy * (unsigned)L.n_init_x;
    int bx = (int)(L.delta_x * ix), ex = (int)(L.delta_x * (ix + 1));
    int by = (int)(L.delta_y * iy), ey = (int)(L.delta_y * (iy + 1));
    uint32_t key = node;
    for (int d = 0; d < kQtDepth; ++d) {
        const int mx = bx + ((ex - bx + 1) >> 1), my = by + ((ey - by + 1) >> 1);
        const unsigned xb = mx <= x, yb = my <= y;
        if (xb) bx = mx; else ex = mx;
        if (yb) by = my; else ey = my;
        key = (key << 2) | (xb | (yb << 1));
    }
    return key;   // node index above 26 digit bits
}

struct QtNode { int s, e, depth; bool leaf; };

// digit of sorted element i at split number `depth` (0 = first split)
inline unsigned qt_digit(uint32_t key, int depth) { return (key >> (2 * (kQtDepth - 1 - depth))) & 3u; }

// children of node n: boundaries b[0..4] (b[0]=s, b[4]=e) of the four digit values
inline void qt_split(const std::vector<uint32_t>& keys, const QtNode& n, int b[5]) {
    b[0] = n.s; b[4] = n.e;
    if (n.depth >= kQtDepth) { b[1] = b[2] = b[3] = n.e; return; }   // unresolvable: stays together (never for distinct pixels)
    for (unsigned v = 1; v <= 3; ++v) {
        int lo = n.s, hi = n.e;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (qt_digit(keys[mid], n.depth) >= v) hi = mid; else lo = mid + 1;
        }
        b[v] = lo;
    }
}

// Returns indices into `c` of the selected candidates, in the reference's output order.
inline std::vector<int> quadtree_select(const QtCand* c, int n, const LevelGeom& L, unsigned N) {
    std::vector<uint32_t> key(n);
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) key[i] = qt_key(c[i].x, c[i].y, L);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key[a] < key[b]; });
    std::vector<uint32_t> skey(n);
    for (int i = 0; i < n; ++i) skey[i] = key[order[i]];

    // initial nodes: ranges of equal node index, in node order; empty ones dropped
    std::vector<QtNode> list;
    {
        int s = 0;
        while (s < n) {
            const uint32_t node = skey[s] >> (2 * kQtDepth);
            int e = s;
            while (e < n && (skey[e] >> (2 * kQtDepth)) == node) ++e;
            list.push_back({s, e, 0, e - s == 1});
            s = e;
        }
    }

    bool filled = false;
    // phase 1: whole-list passes (orb_extractor.cc:482-518)
    while (true) {
        const size_t prev = list.size();
        std::vector<QtNode> created, kept;
        for (const QtNode& nd : list) {
            if (nd.leaf) { kept.push_back(nd); continue; }
            int b[5];
            qt_split(skey, nd, b);
            for (int k = 0; k < 4; ++k)
                if (b[k + 1] > b[k]) created.push_back({b[k], b[k + 1], nd.depth + 1, false});
        }
        size_t pool = 0;
        for (const QtNode& nd : created) pool += (nd.e - nd.s > 1);
        list.assign(created.rbegin(), created.rend());          // push_front order
        list.insert(list.end(), kept.begin(), kept.end());      // flagged leaves keep their place at the tail
        if (N <= list.size() || list.size() == prev) { filled = true; break; }
        if (N < list.size() + pool) break;
    }
    // phase 2: split the fullest nodes first until the quota is met (orb_extractor.cc:520-552)
    while (!filled) {
        const size_t prev = list.size();
        std::vector<int> pool;   // list positions of splittable nodes, (count desc, creation desc) == (count desc, position asc)
        for (int i = 0; i < (int)list.size(); ++i)
            if (!list[i].leaf && list[i].e - list[i].s > 1) pool.push_back(i);
        std::stable_sort(pool.begin(), pool.end(), [&](int a, int b) {
            return list[a].e - list[a].s > list[b].e - list[b].s;
        });
        std::vector<QtNode> created;
        std::vector<char> erased(list.size(), 0);
        size_t size = list.size();
        for (int p : pool) {
            int b[5];
            qt_split(skey, list[p], b);
            int k_nonempty = 0;
            for (int k = 0; k < 4; ++k)
                if (b[k + 1] > b[k]) { created.push_back({b[k], b[k + 1], list[p].depth + 1, false}); ++k_nonempty; }
            erased[p] = 1;
            size += k_nonempty - 1;
            if (N <= size) { filled = true; break; }
        }
        std::vector<QtNode> next(created.rbegin(), created.rend());
        for (size_t i = 0; i < list.size(); ++i)
            if (!erased[i]) next.push_back(list[i]);
        list.swap(next);
        if (filled || N <= list.size() || list.size() == prev) break;
    }
    // first maximum response per node, in node order (orb_extractor.cc:659-685)
    std::vector<int> out;
    out.reserve(list.size());
    for (const QtNode& nd : list) {
        // the reference's node keeps its points in candidate order, so "first maximum" is the
        // maximum score with the smallest candidate index (the sorted range is in key order)
        int best = order[nd.s];
        for (int i = nd.s + 1; i < nd.e; ++i) {
            const int o = order[i];
            if (c[o].score > c[best].score || (c[o].score == c[best].score && o < best)) best = o;
        }
        out.push_back(best);
    }
    return out;
}

}  // namespace plp
