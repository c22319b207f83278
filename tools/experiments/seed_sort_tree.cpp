// Partition tree of the exact seed sort on the replay frames (host model, CPU): how many partitions of which size class a frame has.
//   python tools/experiments/seed_sort_tree_entries.py   (writes /tmp/ss/entries.bin: the seed arrays of 8 replay frames, each followed by its skip key)
//   g++ -O2 -std=c++17 -Istructure-plp-slam_amd/csrc -o /tmp/ss/tree tools/experiments/seed_sort_tree.cpp && /tmp/ss/tree
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include "seed_sort_model.hpp"
using namespace plp::seedsort;
int main() {
    FILE* f = fopen("/tmp/ss/entries.bin", "rb");
    const int n = 76241;
    std::vector<uint32_t> v(n + 1);
    int frame = 0;
    while (fread(v.data(), 4, n + 1, f) == (size_t)n + 1) {
        uint32_t skip = v[n];
        struct Seg { int first, last, depth, level; };
        int lg = 0; while ((2 << lg) <= n) ++lg;
        std::vector<Seg> st{{0, n, 2 * lg, 0}};
        long g_cnt = 0, g_ent = 0, g_swaps = 0, wwg_cnt = 0, wwg_ent = 0, wave_cnt = 0, wave_ent = 0, lane_cnt = 0, lane_ent = 0, uni_cnt = 0, uni_ent = 0, win_cnt = 0, win_ent = 0, skipped_ent = 0;
        long wave_small = 0, wave_small_ent = 0;
        std::map<int, std::pair<long,long>> per_level;
        // window tracking: a segment <= 4096 whose parent > 4096 is a window
        while (!st.empty()) {
            Seg s = st.back(); st.pop_back();
            int ns = s.last - s.first;
            if (ns <= 16) continue;
            bool uniform = true;
            for (int i = s.first + 1; i < s.last && uniform; ++i) uniform = (v[i] >> kKeyShift) == (v[s.first] >> kKeyShift);
            if (uniform && ns <= kUniformMax && uniform_levels(ns) <= s.depth) {
                if ((v[s.first] >> kKeyShift) < skip) { skipped_ent += ns; continue; }
                std::vector<uint32_t> src(v.begin() + s.first, v.begin() + s.last);
                for (int x = s.first; x < s.last; ++x) v[uniform_final_pos(x, s.first, s.last)] = src[x - s.first];
                uni_cnt++; uni_ent += ns; continue;
            }
            if (ns > 4096) { g_cnt++; g_ent += ns; if (frame == 0) printf("  G level %d [%d,%d) n=%d\n", s.level, s.first, s.last, ns); }
            else if (ns > 64) { wave_cnt++; wave_ent += ns; if (ns <= 256) { wave_small++; wave_small_ent += ns; } }
            else { lane_cnt++; lane_ent += ns; }
            per_level[s.level].first++; per_level[s.level].second += ns;
            // count swaps
            std::vector<uint32_t> before(v.begin() + s.first, v.begin() + s.last);
            int cut = partition_model(v.data(), s.first, s.last);
            if (ns > 4096) { long sw = 0; for (int i = 0; i < ns; ++i) sw += before[i] != v[s.first + i]; g_swaps += sw; if (frame == 0) printf("     cut %d moved %ld pivot %u\n", cut, sw, v[s.first] >> kKeyShift); }
            st.push_back({s.first, cut, s.depth - 1, s.level + 1});
            if ((v[s.first] >> kKeyShift) >= skip) st.push_back({cut, s.last, s.depth - 1, s.level + 1});
            else skipped_ent += s.last - cut;
        }
        printf("frame %d: G %ld parts %ld ent moved %ld | wave %ld parts %ld ent (<=256: %ld parts %ld ent) | lane %ld parts %ld ent | uniform %ld segs %ld ent | skipped %ld\n",
               frame, g_cnt, g_ent, g_swaps, wave_cnt, wave_ent, wave_small, wave_small_ent, lane_cnt, lane_ent, uni_cnt, uni_ent, skipped_ent);
        if (frame == 0) for (auto& kv : per_level) printf("   level %d: %ld partitions, %ld entries\n", kv.first, kv.second.first, kv.second.second);
        ++frame;
    }
}
