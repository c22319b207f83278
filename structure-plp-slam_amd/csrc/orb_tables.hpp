// Host-side constant tables of the ORB front-end (product code, uploaded to HBM once per
// image geometry).  Mirrors feature::orb_extractor::initialize / orb_params::calc_*
// (reference src/PLPSLAM/feature/orb_extractor.cc:235-287, orb_params.cc:86-128) and the
// per-level cell grid of compute_fast_keypoints (orb_extractor.cc:338-392).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace plp {

constexpr int kOrbBorder = 19;     // orb_patch_radius_  (orb_extractor.h:148)
constexpr int kFastPatch = 31;     // fast_patch_size_
constexpr int kHalfPatch = 15;     // fast_half_patch_size_
constexpr int kCellSize = 64;      // orb_extractor.cc:339
constexpr int kCellOverlap = 6;    // orb_extractor.cc:338
constexpr int kMaxLevels = 16;
constexpr int kCellCap = 1024;     // NMS maxima cannot be 8-adjacent: <= 32x32 per 64x64 tested block

struct OrbScaleTables {
    std::vector<float> sf, isf, sigma2, isigma2;
    std::vector<uint32_t> quota;
    int u_max[kHalfPatch + 1];
};

inline OrbScaleTables make_scale_tables(uint32_t max_kp, float scale_factor, uint32_t n_levels) {
    OrbScaleTables t;
    t.sf.assign(n_levels, 1.0f); t.isf.assign(n_levels, 1.0f);
    t.sigma2.assign(n_levels, 1.0f); t.isigma2.assign(n_levels, 1.0f);
    float run = 1.0f;
    for (uint32_t l = 1; l < n_levels; ++l) {
        t.sf[l] = scale_factor * t.sf[l - 1];
        t.isf[l] = (1.0f / scale_factor) * t.isf[l - 1];
        run = scale_factor * run;
        t.sigma2[l] = run * run;
        t.isigma2[l] = 1.0f / (run * run);
    }
    // geometric per-level quota, remainder to the top level
    t.quota.assign(n_levels, 0);
    const double inv = 1.0 / scale_factor;
    double want = max_kp * (1.0 - inv) / (1.0 - std::pow(inv, (double)n_levels));
    uint32_t used = 0;
    for (uint32_t l = 0; l + 1 < n_levels; ++l) {
        t.quota[l] = (uint32_t)std::round(want);
        used += t.quota[l];
        want *= inv;
    }
    t.quota[n_levels - 1] = (uint32_t)std::max((int)max_kp - (int)used, 0);
    // half-widths of the radius-15 disc rows
    const double r = kHalfPatch;
    const int vmax = (int)std::floor(r * std::sqrt(2.0) / 2 + 1);
    const int vmin = (int)std::ceil(r * std::sqrt(2.0) / 2);
    for (int v = 0; v <= kHalfPatch; ++v) t.u_max[v] = 0;
    for (int v = 0; v <= vmax; ++v) t.u_max[v] = (int)std::round(std::sqrt(r * r - (double)v * v));
    for (int v = kHalfPatch, v0 = 0; v >= vmin; --v) {
        while (t.u_max[v0] == t.u_max[v0 + 1]) ++v0;
        t.u_max[v] = v0;
        ++v0;
    }
    return t;
}

// cvFloor / cvRound(float) helpers for the table builders
inline int floor_i(float v) { int i = (int)v; return i - (i > v); }
inline short q11(float w) {  // saturate_cast<short>(w * 2048): round-half-even
    const int i = (int)std::nearbyintf(w * 2048);
    return (short)std::min(std::max(i, -32768), 32767);
}

struct LevelGeom {
    int w = 0, h = 0, pitch = 0;   // pitch: bytes per row in the pyramid / blur planes (64-B multiple)
    size_t off = 0;                // byte offset of the level inside one frame's plane set
    // FAST cell grid
    int n_cell_cols = 0, n_cell_rows = 0;   // cells that are actually processed
    int cell_base = 0;                      // index of this level's first cell in the frame's cell list
    int max_bx = 0, max_by = 0;
    int n_init_x = 1, n_init_y = 1;         // quadtree initial grid
    double delta_x = 0, delta_y = 0;
    int sel_cap = 0, sel_base = 0;          // capacity / base slot of the per-level selected list
};

struct CellDesc {       // one FAST work item (uploaded as int4 x 2)
    int32_t level, min_x, min_y, w;   // ROI origin in level pixels, ROI width
    int32_t h, cx, cy, pad;           // ROI height, cell column/row index
};

struct ResizeTables {   // per destination level l>=1, concatenated
    std::vector<int16_t> xofs0, xofs1, a0, a1;     // per dst column
    std::vector<int16_t> yofs0, yofs1, b0, b1;     // per dst row
    std::vector<int> col_base, row_base;           // per level start in the arrays above
};

struct OrbGeometry {
    int rows = 0, cols = 0, n_levels = 0;
    std::vector<LevelGeom> lv;
    std::vector<CellDesc> cells;
    size_t frame_plane_bytes = 0;   // bytes of one frame's pyramid (all levels, padded)
    int total_sel_cap = 0;
    ResizeTables rs;
};

inline OrbGeometry make_geometry(int rows, int cols, const OrbScaleTables& st, int n_levels) {
    OrbGeometry g;
    g.rows = rows; g.cols = cols; g.n_levels = n_levels;
    g.lv.resize(n_levels);
    size_t off = 0;
    for (int l = 0; l < n_levels; ++l) {
        LevelGeom& L = g.lv[l];
        if (l == 0) { L.w = cols; L.h = rows; }
        else {
            const double scale = st.sf[l];
            L.w = (int)std::round(cols * 1.0 / scale);
            L.h = (int)std::round(rows * 1.0 / scale);
        }
        L.pitch = (L.w + 63) / 64 * 64;
        L.off = off;
        off += ((size_t)L.pitch * L.h + 255) / 256 * 256;
    }
    g.frame_plane_bytes = off;

    // resize tables: level l from level l-1
    g.rs.col_base.assign(n_levels, 0); g.rs.row_base.assign(n_levels, 0);
    for (int l = 1; l < n_levels; ++l) {
        const int sw = g.lv[l - 1].w, sh = g.lv[l - 1].h, dw = g.lv[l].w, dh = g.lv[l].h;
        g.rs.col_base[l] = (int)g.rs.xofs0.size();
        g.rs.row_base[l] = (int)g.rs.yofs0.size();
        const double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
        for (int dx = 0; dx < dw; ++dx) {
            float fx = (float)((dx + 0.5) * scale_x - 0.5);
            int sx = floor_i(fx);
            fx -= sx;
            if (sx < 0) { fx = 0; sx = 0; }
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
            g.rs.xofs0.push_back((int16_t)sx);
            g.rs.xofs1.push_back((int16_t)std::min(sx + 1, sw - 1));
            g.rs.a0.push_back(q11(1.f - fx));
            g.rs.a1.push_back(q11(fx));
        }
        for (int dy = 0; dy < dh; ++dy) {
            float fy = (float)((dy + 0.5) * scale_y - 0.5);
            int sy = floor_i(fy);
            fy -= sy;
            g.rs.yofs0.push_back((int16_t)std::min(std::max(sy, 0), sh - 1));
            g.rs.yofs1.push_back((int16_t)std::min(std::max(sy + 1, 0), sh - 1));
            g.rs.b0.push_back(q11(1.f - fy));
            g.rs.b1.push_back(q11(fy));
        }
    }

    // FAST cells, reference iteration order (level, cell row i, cell col j)
    int sel_base = 0;
    for (int l = 0; l < n_levels; ++l) {
        LevelGeom& L = g.lv[l];
        L.cell_base = (int)g.cells.size();
        L.max_bx = L.w - kOrbBorder; L.max_by = L.h - kOrbBorder;
        const int width = L.max_bx - kOrbBorder, height = L.max_by - kOrbBorder;
        L.sel_cap = 2 * (int)st.quota[l] + 4;   // a whole-list pass may overshoot the quota, but never beyond 2x
        L.sel_base = sel_base;
        sel_base += L.sel_cap;
        if (width <= 0 || height <= 0) continue;
        const int num_cols = width / kCellSize + 1, num_rows = height / kCellSize + 1;
        int n_rows_used = 0, n_cols_used = 0;
        for (int i = 0; i < num_rows; ++i) {
            const int min_y = kOrbBorder + i * kCellSize;
            if (L.max_by - kCellOverlap <= min_y) continue;
            const int max_y = std::min(min_y + kCellSize + kCellOverlap, L.max_by);
            ++n_rows_used;
            n_cols_used = 0;
            for (int j = 0; j < num_cols; ++j) {
                const int min_x = kOrbBorder + j * kCellSize;
                if (L.max_bx - kCellOverlap <= min_x) continue;
                const int max_x = std::min(min_x + kCellSize + kCellOverlap, L.max_bx);
                ++n_cols_used;
                g.cells.push_back({l, min_x, min_y, max_x - min_x, max_y - min_y, j, i, 0});
            }
        }
        L.n_cell_rows = n_rows_used; L.n_cell_cols = n_cols_used;
        // quadtree initial grid (orb_extractor.cc:561-582), border-relative frame
        const int min_x = kOrbBorder, max_x = L.max_bx, min_y = kOrbBorder, max_y = L.max_by;
        const double ratio = (double)(max_x - min_x) / (max_y - min_y);
        if (ratio > 1) {
            L.n_init_x = (int)std::round(ratio); L.n_init_y = 1;
            L.delta_x = (double)(max_x - min_x) / L.n_init_x;
            L.delta_y = max_y - min_y;
        } else {
            L.n_init_x = 1; L.n_init_y = (int)std::round(1 / ratio);
            L.delta_x = max_x - min_y;   // reference quirk kept (orb_extractor.cc:580)
            L.delta_y = (double)(max_y - min_y) / L.n_init_y;
        }
    }
    g.total_sel_cap = sel_base;
    return g;
}

}  // namespace plp
