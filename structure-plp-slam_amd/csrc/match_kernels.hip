// Hand-written HIP kernels (gfx950, wave64) of the Hamming matchers, array form.
//
// The reference's point matchers (src/PLPSLAM/match/projection.cc:37-121, :214-358 and
// robust.cc:257-385) are sequential loops: each query (landmark / last-frame key point /
// key-frame key point) takes the best still-free key point in its candidate set, and a taken
// key point is skipped by every later query.  Two steps reproduce that exactly: a top-K candidate list per query
// (independent of the other queries), then the resolution of the claims in query order.
//
//   k_match_prep        per frame: free in-grid targets counting-sorted by (grid column, grid row, index) = the visiting
//                       order of data::get_keypoints_in_cell (common.cc:241-313), 16-byte records + cell starts
//   k_match_topk_cells  windowed modes, main path: 16 lanes per query walk the window's grid columns staged in LDS
//                       (12-byte records), keep the K = 8 best by (distance, visiting order)
//   k_match_topk_lds    brute-force mode: the target descriptors staged in LDS, one wave per query
//   k_match_topk        generic path (line modes, BoW / triangulation groups, frames beyond the LDS budget): one wave per
//                       query scans all targets through candidate_key(); waves stride over a frame's queries
//   k_match_topk_lanes  the same search, one LANE per query: small target sets (key lines) in large batches
//   k_match_resolve     one workgroup per problem: chunk-wise fixed point of "best free candidate given the claims of
//                       all EARLIER queries" (2-3 iterations per chunk of 256 queries), an exact wave-wide rescan for a
//                       query whose truncated list ran dry, then the accept rules, the delta-angle histogram check
//                       (match/angle_checker.h) and the scatter of the results
//   k_match_fuse        independent searches (fuse.cc, projection.cc:894-1142): best candidate per query, no claims
//   k_match_area        area::match_in_consistent_area (area.cc:33-153), one wave per problem
//   k_lbd_match_1nn     BinaryDescriptorMatcher::match: exact 1-NN with MIH discovery-order ties
//   k_hamming_matrix    full nq x nt distance matrix, 64x64 tiles staged through LDS
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>

#include "libstdcxx_sort.hpp"
#include "plp_barrier.hpp"
#include "match_device.hpp"
#include "plp_common.hpp"
#include "xcd_map.hpp"

namespace plp {

__device__ __forceinline__ int floor_d(double v) { int i = (int)v; return i - (i > v); }   // cvFloor(double)
__device__ __forceinline__ int ceil_d(double v) { int i = (int)v; return i + (i < v); }    // cvCeil(double)

__device__ __forceinline__ unsigned hamming256(const uint4& a0, const uint4& a1, const uint4& b0, const uint4& b1) {
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, o), hi = __shfl_xor((unsigned)(v >> 32), o);
        const unsigned long long w = ((unsigned long long)hi << 32) | lo;
        v = w < v ? w : v;
    }
    return v;
}

// Candidate filter + key of (query q, target t); returns ~0ull when t is not a candidate.
// key = distance << 32 | visiting order << 4 | octave   (order = grid cell (col-major) then index)
struct QueryCtx {
    float rx, ry, mg, xr, xr2;
    int min_level, max_level, min_cx, max_cx, min_cy, max_cy;
    double l0, l1, l2, lden;   // line modes: the projected line (sp x ep) and sqrt(l0^2 + l1^2)
    double rdx, rdy;           // FUSE: f64 reprojection
    int group;                 // BOW: node id of the query feature
    unsigned pred;             // FUSE: predicted scale level (unsigned, as the reference)
    double g0, g1, g2, gden;   // FUSE_LINE: the projected line from the f64 reprojections (chi-square gate)
    double b1x, b1y, b1z, thr; // TRIANGULATION: bearing of the query, residual threshold
    bool stereo;
    bool windowed, empty, line;
};
__device__ __forceinline__ bool is_line_mode(int mode) { return mode == PLP_MATCH_MODE_LANDMARKS_LINE || mode == PLP_MATCH_MODE_LAST_FRAME_LINE || mode == PLP_MATCH_MODE_FUSE_LINE; }
__device__ __forceinline__ bool is_group_mode(int mode) { return mode == PLP_MATCH_MODE_BOW || mode == PLP_MATCH_MODE_TRIANGULATION; }
__device__ __forceinline__ bool is_last_frame_mode(int mode) { return mode == PLP_MATCH_MODE_LAST_FRAME || mode == PLP_MATCH_MODE_LAST_FRAME_LINE; }

// The generic kernels (k_match_topk, k_match_topk_lanes, k_match_resolve_generic) are instantiated per FAMILY of modes: written against
// every mode at once they kept all modes' pointers and gates live (81-125 scalar registers spilled, a scratch frame per wave).  An
// instantiation reads the mode through fam_mode<FAM>(): a select over the constants of its family, so every test for another family's
// mode folds away at compile time and that family's arguments are never loaded.  launch_match() picks the instantiation from the same value.
enum MatchFamily { kFamAny = 0, kFamLine = 1, kFamGroup = 2, kFamPoint = 3, kFamGrid = 4 };   // kFamGrid: the two windowed point modes after k_match_prep
template <int FAM>
__device__ __forceinline__ int fam_mode(int m) {
    if constexpr (FAM == kFamLine) return m == PLP_MATCH_MODE_LANDMARKS_LINE ? PLP_MATCH_MODE_LANDMARKS_LINE : PLP_MATCH_MODE_LAST_FRAME_LINE;
    else if constexpr (FAM == kFamGroup) return m == PLP_MATCH_MODE_BOW ? PLP_MATCH_MODE_BOW : PLP_MATCH_MODE_TRIANGULATION;
    else if constexpr (FAM == kFamPoint) return m == PLP_MATCH_MODE_LANDMARKS ? PLP_MATCH_MODE_LANDMARKS : m == PLP_MATCH_MODE_LAST_FRAME ? PLP_MATCH_MODE_LAST_FRAME : PLP_MATCH_MODE_BRUTE_FORCE;
    else if constexpr (FAM == kFamGrid) return m == PLP_MATCH_MODE_LANDMARKS ? PLP_MATCH_MODE_LANDMARKS : PLP_MATCH_MODE_LAST_FRAME;
    else return m;
}

template <int FAM = kFamAny>
__device__ __forceinline__ QueryCtx make_query(const MatchProblem& P, int q, int b) {
    const int mode = fam_mode<FAM>(P.mode);
    const size_t qoff = (size_t)b * P.m_cap;
    const float* reproj = P.q_reproj ? P.q_reproj + qoff * 2 : nullptr;
    const float* q_xr = P.q_x_right ? P.q_x_right + qoff : nullptr;
    const int32_t* q_level = P.q_level ? P.q_level + qoff : nullptr;
    QueryCtx c{};
    c.windowed = mode != PLP_MATCH_MODE_BRUTE_FORCE && !is_group_mode(mode);
    c.line = is_line_mode(mode);
    if (is_group_mode(mode)) c.group = P.q_group[qoff + q];
    if (mode == PLP_MATCH_MODE_TRIANGULATION) {
        const double* bq = P.q_bearing + (qoff + q) * 3;
        c.b1x = bq[0]; c.b1y = bq[1]; c.b1z = bq[2];
        c.stereo = q_xr && 0 <= q_xr[q];
        // residual_rad_thr * bearing_1_scale_factor (robust.cc:399-404)
        c.thr = (0.2 * 3.14159265358979323846 / 180.0) * (double)P.scale_factors[q_level[q] & 15];
    }
    if (!c.windowed) return c;
    const int lvl = q_level[q];
    // the reference indexes scale_factors_.at(level) and throws on a level outside the table; device arrays cannot be validated by
    // the host entry, so the table lookup is clamped (the level WINDOW below still uses the caller's value)
    const int lvl_tab = min(max(lvl, 0), max(P.num_levels, 1) - 1);
    if (c.line) {   // data::get_keylines_in_cell (common.cc:315-363) + the level windows of projection.cc:138-144, :429-450
        c.mg = __fmul_rn(P.margin, P.scale_factors[lvl_tab]);
        double x1, y1, x2, y2;
        if (mode == PLP_MATCH_MODE_FUSE_LINE) {   // the f64 reprojections are narrowed to float by the call (fuse.cc:420-422)
            const double* a = P.q_reproj_d + (qoff + q) * 2; const double* e = P.q_reproj2_d + (qoff + q) * 2;
            x1 = (float)a[0]; y1 = (float)a[1]; x2 = (float)e[0]; y2 = (float)e[1];
            c.g0 = a[1] * 1.0 - 1.0 * e[1]; c.g1 = 1.0 * e[0] - a[0] * 1.0; c.g2 = a[0] * e[1] - a[1] * e[0];
            c.gden = sqrt(c.g0 * c.g0 + c.g1 * c.g1);
        } else {
            const float* r2 = P.q_reproj2 + qoff * 2;
            x1 = reproj[2 * q]; y1 = reproj[2 * q + 1]; x2 = r2[2 * q]; y2 = r2[2 * q + 1];
        }
        c.l0 = y1 * 1.0 - 1.0 * y2; c.l1 = 1.0 * x2 - x1 * 1.0; c.l2 = x1 * y2 - y1 * x2;
        c.lden = sqrt(c.l0 * c.l0 + c.l1 * c.l1);
        c.xr = q_xr ? q_xr[q] : -1.f;
        c.xr2 = P.q_x_right2 ? P.q_x_right2[qoff + q] : -1.f;
        if (mode == PLP_MATCH_MODE_FUSE_LINE) { c.min_level = -1; c.max_level = -1; }
        else if (mode == PLP_MATCH_MODE_LANDMARKS_LINE || P.level_window == 1) { c.min_level = lvl - 1; c.max_level = lvl; }
        else if (P.level_window == 2) { c.min_level = lvl - 1; c.max_level = lvl + 1; }
        else if (P.direction == 1) { c.min_level = lvl; c.max_level = P.num_levels_lsd; }
        else if (P.direction == 2) { c.min_level = 0; c.max_level = lvl + 1; }
        else { c.min_level = lvl - 1; c.max_level = lvl + 1; }
        c.empty = false;
        return c;
    }
    if (mode == PLP_MATCH_MODE_FUSE) {   // get_keypoints_in_cell(reproj(0), reproj(1), margin * scale) takes the f64 reprojection as float
        const double* rd = P.q_reproj_d + qoff * 2;
        c.rdx = rd[2 * q]; c.rdy = rd[2 * q + 1];
        c.rx = (float)c.rdx; c.ry = (float)c.rdy;
        c.pred = (unsigned)lvl;
    } else { c.rx = reproj[2 * q]; c.ry = reproj[2 * q + 1]; }
    c.mg = __fmul_rn(P.margin, P.scale_factors[lvl_tab]);
    c.xr = q_xr ? q_xr[q] : -1.f;
    if (mode == PLP_MATCH_MODE_LANDMARKS || (mode == PLP_MATCH_MODE_LAST_FRAME && P.level_window == 1)) { c.min_level = lvl - 1; c.max_level = lvl; }
    else if (mode == PLP_MATCH_MODE_FUSE) { c.min_level = -1; c.max_level = -1; }
    else if (P.level_window == 2) { c.min_level = lvl - 1; c.max_level = lvl + 1; }
    else if (P.direction == 1) { c.min_level = lvl; c.max_level = P.num_levels - 1; }
    else if (P.direction == 2) { c.min_level = 0; c.max_level = lvl; }
    else { c.min_level = lvl - 1; c.max_level = lvl + 1; }
    // cell range of the window (common.cc:249-271)
    // float window arithmetic, double cell scale (inv_cell_width_ is a double)
    c.min_cx = max(0, floor_d((double)__fsub_rn(__fsub_rn(c.rx, P.grid_min_x), c.mg) * P.inv_cell_w));
    c.max_cx = min(P.grid_cols - 1, ceil_d((double)__fadd_rn(__fsub_rn(c.rx, P.grid_min_x), c.mg) * P.inv_cell_w));
    c.min_cy = max(0, floor_d((double)__fsub_rn(__fsub_rn(c.ry, P.grid_min_y), c.mg) * P.inv_cell_h));
    c.max_cy = min(P.grid_rows - 1, ceil_d((double)__fadd_rn(__fsub_rn(c.ry, P.grid_min_y), c.mg) * P.inv_cell_h));
    c.empty = P.grid_cols <= c.min_cx || c.max_cx < 0 || P.grid_rows <= c.min_cy || c.max_cy < 0;
    // match_by_Sim3_transform tests `scale_level < pred - 1 || pred < scale_level` on unsigned values (projection.cc:862)
    if (mode == PLP_MATCH_MODE_LAST_FRAME && (P.flags & PLP_MATCH_FLAG_UNSIGNED_LEVEL) && lvl == 0) c.empty = true;
    return c;
}

// What the two projection line matchers test of a target key line (data::get_keylines_in_cell, common.cc:315-363, + projection.cc:138-144,
// :429-450), on the target's fields alone: candidate_key() fills them from memory, k_match_topk_lanes from a register broadcast.
struct LineTarget { float sx, sy, ex, ey; int octave; bool occ; float xr, xr2; unsigned kp_oct; };
template <int FAM = kFamAny>
__device__ __forceinline__ bool line_gate(const MatchProblem& P, const QueryCtx& c, const LineTarget& T, bool has_xr) {
    const int mode = fam_mode<FAM>(P.mode);
    const float dsp = (float)(((double)T.sx * c.l0 + (double)T.sy * c.l1 + c.l2) / c.lden);
    const float dep = (float)(((double)T.ex * c.l0 + (double)T.ey * c.l1 + c.l2) / c.lden);
    if (fabsf(dsp) > c.mg || fabsf(dep) > c.mg) return false;
    const bool check_level = (0 < c.min_level) || (0 <= c.max_level);
    if (check_level) {
        if (T.octave < c.min_level) return false;
        if (c.max_level > 0 && T.octave > c.max_level) return false;   // `max_level > 0` as the reference (common.cc:354)
    }
    if (T.occ) return false;
    if (mode == PLP_MATCH_MODE_LAST_FRAME_LINE && P.is_rgbd && has_xr && P.t_x_right2) {
        const float a = T.xr, b2 = T.xr2;
        if (a > 0 && b2 > 0 && (c.mg < fabsf(__fsub_rn(c.xr, a)) || c.mg < fabsf(__fsub_rn(c.xr2, b2)))) return false;
    }
    return true;
}

template <int FAM = kFamAny>
__device__ __forceinline__ unsigned long long candidate_key(const MatchProblem& P, const QueryCtx& c, int t, const plp_keypoint* kps,
                                                           const uint8_t* t_desc, const float* t_xr, const uint8_t* t_occ,
                                                           const uint4& q0, const uint4& q1) {
    const int mode = fam_mode<FAM>(P.mode);
    unsigned order = (unsigned)t, oct = 0;
    if (is_group_mode(mode)) {
        const size_t tb = (size_t)(t_desc - P.t_desc) / 32;
        if (P.t_group[tb + t] != c.group) return ~0ull;
        if (t_occ && t_occ[t]) return ~0ull;
        if (mode == PLP_MATCH_MODE_TRIANGULATION) {
            const uint4* d = reinterpret_cast<const uint4*>(t_desc + 32 * (size_t)t);
            const unsigned dist = hamming256(q0, q1, d[0], d[1]);
            if (50u < dist) return ~0ull;                                       // HAMMING_DIST_THR_LOW (robust.cc:124)
            const double* b2 = P.t_bearing + (tb + t) * 3;
            const double* E = P.epipolar + (tb / P.n_cap) * 12;
            const bool stereo2 = t_xr && 0 <= t_xr[t];
            if (!c.stereo && !stereo2) {                                        // not near the epipole (:129-141)
                const double cos_dist = E[9] * b2[0] + E[10] * b2[1] + E[11] * b2[2];
                if (0.99862953475 < cos_dist) return ~0ull;
            }
            // check_epipolar_constraint (:387-405)
            const double n0 = E[0] * b2[0] + E[1] * b2[1] + E[2] * b2[2], n1 = E[3] * b2[0] + E[4] * b2[1] + E[5] * b2[2],
                         n2 = E[6] * b2[0] + E[7] * b2[1] + E[8] * b2[2];
            const double cos_residual = (n0 * c.b1x + n1 * c.b1y + n2 * c.b1z) / sqrt(n0 * n0 + n1 * n1 + n2 * n2);
            const double residual_rad = 3.14159265358979323846 / 2.0 - fabs(acos(cos_residual));
            if (!(residual_rad < c.thr)) return ~0ull;
            return ((unsigned long long)dist << 32) | ((unsigned long long)(0xffffu - (unsigned)t) << 4);   // ties: the later candidate
        }
    } else if (c.line && mode != PLP_MATCH_MODE_FUSE_LINE) {
        const size_t tb = (size_t)(t_desc - P.t_desc) / 32;   // per-problem target offset
        const plp_keyline kl = P.t_kl[tb + t];
        LineTarget T;
        T.sx = kl.startPointX; T.sy = kl.startPointY; T.ex = kl.endPointX; T.ey = kl.endPointY; T.octave = kl.octave;
        T.occ = t_occ && t_occ[t];
        T.xr = t_xr ? t_xr[t] : 0.f; T.xr2 = P.t_x_right2 ? P.t_x_right2[tb + t] : 0.f;
        T.kp_oct = P.t_kp_octave ? ((unsigned)P.t_kp_octave[tb + t] & 15u) : 0u;
        if (!line_gate<FAM>(P, c, T, t_xr != nullptr)) return ~0ull;
        oct = T.kp_oct;
    } else if (c.line) {
        const size_t tb = (size_t)(t_desc - P.t_desc) / 32;   // per-problem target offset
        const plp_keyline kl = P.t_kl[tb + t];
        const float dsp = (float)(((double)kl.startPointX * c.l0 + (double)kl.startPointY * c.l1 + c.l2) / c.lden);
        const float dep = (float)(((double)kl.endPointX * c.l0 + (double)kl.endPointY * c.l1 + c.l2) / c.lden);
        if (fabsf(dsp) > c.mg || fabsf(dep) > c.mg) return ~0ull;
        {   // fuse.cc:440-451, f64
            const double e_sp = ((double)kl.startPointX * c.g0 + (double)kl.startPointY * c.g1 + c.g2) / c.gden;
            const double e_ep = ((double)kl.endPointX * c.g0 + (double)kl.endPointY * c.g1 + c.g2) / c.gden;
            if ((double)5.99146f < (e_sp * e_sp + e_ep * e_ep) * (double)P.inv_level_sigma_sq[(unsigned)kl.octave & 15]) return ~0ull;
            const uint4* d = reinterpret_cast<const uint4*>(t_desc + 32 * (size_t)t);
            return ((unsigned long long)hamming256(q0, q1, d[0], d[1]) << 32) | ((unsigned long long)(unsigned)t << 4);
        }
    } else if (c.windowed) {
        const plp_keypoint k = kps[t];
        const int cx = floor_d((double)__fsub_rn(k.x, P.grid_min_x) * P.inv_cell_w);
        const int cy = floor_d((double)__fsub_rn(k.y, P.grid_min_y) * P.inv_cell_h);
        if (cx < 0 || cx >= P.grid_cols || cy < 0 || cy >= P.grid_rows) return ~0ull;        // not in the grid at all
        if (cx < c.min_cx || cx > c.max_cx || cy < c.min_cy || cy > c.max_cy) return ~0ull;
        const bool check_level = (0 < c.min_level) || (0 <= c.max_level);
        if (check_level) {
            if (k.octave < c.min_level) return ~0ull;
            if (0 <= c.max_level && c.max_level < k.octave) return ~0ull;
        }
        if (!(fabsf(__fsub_rn(k.x, c.rx)) < c.mg && fabsf(__fsub_rn(k.y, c.ry)) < c.mg)) return ~0ull;
        if (mode == PLP_MATCH_MODE_FUSE) {   // fuse.cc:230-262: octave window in unsigned arithmetic + chi-square gates in f64
            const unsigned sl = (unsigned)k.octave;
            if (P.flags & PLP_MATCH_FLAG_SIGNED_LEVEL) { if (k.octave < (int)c.pred - 1 || (int)c.pred < k.octave) return ~0ull; }
            else if (sl < c.pred - 1u || c.pred < sl) return ~0ull;
            if (!(P.flags & PLP_MATCH_FLAG_NO_CHI2)) {
            const double e_x = c.rdx - (double)k.x, e_y = c.rdy - (double)k.y;
            const float xr = t_xr ? t_xr[t] : -1.f;
            if (xr >= 0) {
                const float e_xr = __fsub_rn(c.xr, xr);
                const double err = e_x * e_x + e_y * e_y + (double)__fmul_rn(e_xr, e_xr);
                if ((double)7.81473f < err * (double)P.inv_level_sigma_sq[sl & 15]) return ~0ull;
            } else {
                const double err = e_x * e_x + e_y * e_y;
                if ((double)5.99146f < err * (double)P.inv_level_sigma_sq[sl & 15]) return ~0ull;
            }
            }
        } else {
        if (t_occ && t_occ[t]) return ~0ull;                                                 // already holds an observed landmark
        if (t_xr) {
            const float xr = t_xr[t];
            if (0 < xr && c.mg < fabsf(__fsub_rn(c.xr, xr))) return ~0ull;                    // stereo gate
        }
        }
        order = ((unsigned)(cx * P.grid_rows + cy) << 16) | (unsigned)t;
        oct = (unsigned)k.octave & 15u;
    }
    const uint4* d = reinterpret_cast<const uint4*>(t_desc + 32 * (size_t)t);
    const unsigned dist = hamming256(q0, q1, d[0], d[1]);
    return ((unsigned long long)dist << 32) | ((unsigned long long)order << 4) | oct;
}

// lane-local sorted insertion + wave merge of the K best keys of one query
__device__ __forceinline__ void topk_insert(unsigned long long (&top)[kMatchK], unsigned long long key) {
    if (key < top[kMatchK - 1]) {
        top[kMatchK - 1] = key;
#pragma unroll
        for (int i = kMatchK - 1; i > 0; --i)
            if (top[i] < top[i - 1]) { const unsigned long long s = top[i]; top[i] = top[i - 1]; top[i - 1] = s; }
    }
}
// the merged list is stored packed: distance << 20 | octave << 16 | target index (0xffffffff = none), 32 B per query
__device__ __forceinline__ uint32_t pack_key(unsigned long long key) {
    return key == ~0ull ? 0xffffffffu : ((uint32_t)(key >> 32) << 20) | ((uint32_t)(key & 15) << 16) | (uint32_t)((key >> 4) & 0xffff);
}
__device__ __forceinline__ void topk_merge_store(unsigned long long (&top)[kMatchK], int lane, uint32_t* klist) {
#pragma unroll
    for (int r = 0; r < kMatchK; ++r) {
        const unsigned long long mn = wave_min_u64(top[0]);
        if (lane == 0) klist[r] = pack_key(mn);
        if (top[0] == mn && mn != ~0ull) {
#pragma unroll
            for (int i = 0; i + 1 < kMatchK; ++i) top[i] = top[i + 1];
            top[kMatchK - 1] = ~0ull;
        }
    }
}

// Fallback for frames with more key points than the LDS staging of k_match_topk_lds holds:
// grid = (ceil(m_cap / 4), B), block = 256: one wave per query, targets read from HBM/L2.
template <int FAM = kFamAny>
__device__ __forceinline__ void match_topk_query(const MatchProblem& P, int b, int q, int lane) {
    uint32_t* klist = P.klist + ((size_t)b * P.m_cap + q) * kMatchK;
    int32_t* kcount = P.kcount + (size_t)b * P.m_cap + q;
    const uint8_t* q_valid = P.q_valid ? P.q_valid + (size_t)b * P.m_cap : nullptr;
    if (q_valid && !q_valid[q]) { if (lane == 0) *kcount = -1; return; }
    const int n = P.t_counts ? min(P.t_counts[b], P.n_cap) : P.n_cap;
    const plp_keypoint* kps = P.t_kps ? P.t_kps + (size_t)b * P.n_cap : nullptr;
    const uint8_t* t_desc = P.t_desc + (size_t)b * P.n_cap * 32;
    const float* t_xr = P.t_x_right ? P.t_x_right + (size_t)b * P.n_cap : nullptr;
    const uint8_t* t_occ = P.t_occupied ? P.t_occupied + (size_t)b * P.n_cap : nullptr;
    const QueryCtx c = make_query<FAM>(P, q, b);
    const uint4* qd = reinterpret_cast<const uint4*>(P.q_desc + ((size_t)b * P.q_desc_stride + q) * 32);
    const uint4 q0 = qd[0], q1 = qd[1];
    unsigned long long top[kMatchK];
#pragma unroll
    for (int i = 0; i < kMatchK; ++i) top[i] = ~0ull;
    int passed = 0;
    if (!(c.windowed && c.empty))
        for (int t = lane; t < n; t += 64) {
            const unsigned long long key = candidate_key<FAM>(P, c, t, kps, t_desc, t_xr, t_occ, q0, q1);
            if (key == ~0ull) continue;
            ++passed;
            topk_insert(top, key);
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) passed += __shfl_xor(passed, o);
    topk_merge_store(top, lane, klist);
    if (lane == 0) *kcount = passed;
}

// grid = (gx, B) with gx <= ceil(m_cap / 4): the waves of a frame stride over its queries, so a batch of frames with few
// queries each (key lines: ~50 of a 512 capacity) does not launch hundreds of thousands of workgroups that only exit.
template <int FAM>
__global__ __launch_bounds__(256) void k_match_topk(MatchProblem P) {
    const int lane = threadIdx.x & 63, b = blockIdx.y;
    const int m = P.q_counts ? min(P.q_counts[b], P.m_cap) : P.m_cap;
    for (int q = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); q < m; q += gridDim.x * 4) match_topk_query<FAM>(P, b, q, lane);   // the wave's query index: scalar
}

// The same search with ONE LANE per query (each lane scans all targets and keeps its 8 best keys): for small target sets -- the key
// lines of a frame, ~50 against ~50 -- a wave per query leaves most lanes without a target and spends more on merging the lanes'
// lists than on the candidates.  grid = (min(2, ceil(m_cap / 64)), B), block = 64: the one or two waves of a frame walk its blocks of 64
// queries (a frame has ~50-100 of a capacity of 512-1024: a workgroup per BLOCK of the capacity launched 24 576 waves per call of which
// 20 000 only found that their block was empty, and the kernel took 0.28 ms for 60 us of work per active wave).
template <int FAM>
__device__ __forceinline__ void match_topk_lanes_block(const MatchProblem& P, int b, int qblock, int m) {
    const int q = qblock * 64 + (int)threadIdx.x;
    // a lane without a query stays (the line family's loop below has every lane fetch one TARGET per chunk): `active` gates its query work
    bool active = q < m;
    uint32_t* klist = P.klist + ((size_t)b * P.m_cap + q) * kMatchK;
    int32_t* kcount = P.kcount + (size_t)b * P.m_cap + q;
    const uint8_t* q_valid = P.q_valid ? P.q_valid + (size_t)b * P.m_cap : nullptr;
    if (active && q_valid && !q_valid[q]) { *kcount = -1; active = false; }
    if constexpr (FAM != kFamLine) { if (!active) return; }
    const int n = P.t_counts ? min(P.t_counts[b], P.n_cap) : P.n_cap;
    const plp_keypoint* kps = P.t_kps ? P.t_kps + (size_t)b * P.n_cap : nullptr;
    const uint8_t* t_desc = P.t_desc + (size_t)b * P.n_cap * 32;
    const float* t_xr = P.t_x_right ? P.t_x_right + (size_t)b * P.n_cap : nullptr;
    const uint8_t* t_occ = P.t_occupied ? P.t_occupied + (size_t)b * P.n_cap : nullptr;
    QueryCtx c{};
    uint4 q0 = make_uint4(0, 0, 0, 0), q1 = q0;
    if (active) {
        c = make_query<FAM>(P, q, b);
        const uint4* qd = reinterpret_cast<const uint4*>(P.q_desc + ((size_t)b * P.q_desc_stride + q) * 32);
        q0 = qd[0]; q1 = qd[1];
    }
    unsigned long long top[kMatchK];
#pragma unroll
    for (int i = 0; i < kMatchK; ++i) top[i] = ~0ull;
    int passed = 0;
    if constexpr (FAM == kFamLine) {
        // The gates read ~9 fields of the target key line: fetched inside the loop they were five DEPENDENT scalar round trips per target
        // (start point -> end point -> octave -> occupancy -> descriptor ...), ~2 us each beside a busy chip.  Here lane j fetches the
        // fields of target base + j once (all loads of a lane in flight together), and the loop over the targets hands them to all lanes
        // with v_readlane (t is uniform): no memory access per target except the descriptor of one that passed the gates.
        const size_t tb = (size_t)b * P.n_cap;
        const int lane = threadIdx.x;
        for (int base = 0; base < n; base += 64) {
            const int tj = base + lane;
            float f_sx = 0.f, f_sy = 0.f, f_ex = 0.f, f_ey = 0.f, f_xr = 0.f, f_xr2 = 0.f;
            int f_misc = 0;                                   // octave << 8 | occupied << 4 | key-point octave
            if (tj < n) {
                const plp_keyline* kl = P.t_kl + tb + tj;
                f_sx = kl->startPointX; f_sy = kl->startPointY; f_ex = kl->endPointX; f_ey = kl->endPointY;
                const int occ = t_occ && t_occ[tj];
                const unsigned kpo = P.t_kp_octave ? ((unsigned)P.t_kp_octave[tb + tj] & 15u) : 0u;
                f_misc = (int)(((unsigned)kl->octave << 8) | ((unsigned)occ << 4) | kpo);
                if (t_xr) f_xr = t_xr[tj];
                if (P.t_x_right2) f_xr2 = P.t_x_right2[tb + tj];
            }
            const int m_chunk = min(64, n - base);
            for (int u = 0; u < m_chunk; ++u) {
                auto bc_f = [&](float v) -> float { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), u)); };
                LineTarget T;
                T.sx = bc_f(f_sx); T.sy = bc_f(f_sy); T.ex = bc_f(f_ex); T.ey = bc_f(f_ey); T.xr = bc_f(f_xr); T.xr2 = bc_f(f_xr2);
                const int misc = __builtin_amdgcn_readlane(f_misc, u);
                T.octave = misc >> 8; T.occ = (misc >> 4) & 1; T.kp_oct = (unsigned)misc & 15u;   // octave: arithmetic shift keeps the sign
                if (!active || !line_gate<FAM>(P, c, T, t_xr != nullptr)) continue;
                const int t = base + u;
                const uint4* d = reinterpret_cast<const uint4*>(t_desc + 32 * (size_t)t);
                const unsigned dist = hamming256(q0, q1, d[0], d[1]);
                ++passed;
                topk_insert(top, ((unsigned long long)dist << 32) | ((unsigned long long)(unsigned)t << 4) | T.kp_oct);
            }
        }
    } else if (!(c.windowed && c.empty))
        for (int t = 0; t < n; ++t) {
            const unsigned long long key = candidate_key<FAM>(P, c, t, kps, t_desc, t_xr, t_occ, q0, q1);
            if (key == ~0ull) continue;
            ++passed;
            topk_insert(top, key);
        }
    if (!active) return;
#pragma unroll
    for (int r = 0; r < kMatchK; ++r) klist[r] = pack_key(top[r]);
    *kcount = passed;
}
template <int FAM>
__global__ __launch_bounds__(64) void k_match_topk_lanes(MatchProblem P) {
    const int b = blockIdx.y;
    const int m = P.q_counts ? min(P.q_counts[b], P.m_cap) : P.m_cap;
    for (int qblock = blockIdx.x; qblock * 64 < m; qblock += gridDim.x) match_topk_lanes_block<FAM>(P, b, qblock, m);
}

// Main path.  k_match_prep buckets each frame's free, in-grid targets by grid ROW once (counting sort with LDS
// atomics; the order inside a bucket is irrelevant because every candidate key carries the reference's visiting
// order) into 16-byte records in HBM; k_match_topk_lds copies that array into LDS once per workgroup and reuses it
// for kQueriesPerBlock queries, each of which only walks the rows its window overlaps.  (A wave scanning 1000
// key points from L2 for each of ~3000 queries had made the first version L2-bandwidth bound.)
//   windowed modes: {x, y, octave | cell col | cell row, index} (+4 B stereo x_right)
//   brute force:    the 32-byte descriptors themselves are staged
constexpr int kQueriesPerBlock = 128;
constexpr int kCellStride = 4104;  // cell_start[cols * rows + 1] per frame (<= 4097 entries), padded

// k_match_prep: the free, in-grid targets of a frame sorted by (grid column, grid row, index) -- the order in which
// data::get_keypoints_in_cell visits them (common.cc:271-309) -- so that a candidate's rank in the reference's visiting
// order is simply its position in this array (16 bits), and a window is one contiguous range per grid column.
// grid = (B), block = 256.
__global__ __launch_bounds__(256) void k_match_prep(MatchProblem P) {
    // 33 KB of LDS: two of these workgroups fit beside the 85 KB that region growing holds on a CU (with 32-bit cell counters it was 41 KB and
    // one).  Cell counts (at most 8192 targets per frame) are 16-bit halves of 32-bit words: cell c lives in word c >> 1, half c & 1.
    __shared__ uint32_t cnt2[2048];
    __shared__ uint16_t start[4098];
    __shared__ uint16_t tmp_t[8192];
    __shared__ int part[4];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int n = P.t_counts ? min(P.t_counts[b], P.n_cap) : P.n_cap;
    const int ncell = P.grid_cols * P.grid_rows;
    const plp_keypoint* kps = P.t_kps + (size_t)b * P.n_cap;
    const float* t_xr = P.t_x_right ? P.t_x_right + (size_t)b * P.n_cap : nullptr;
    const uint8_t* t_occ = P.t_occupied ? P.t_occupied + (size_t)b * P.n_cap : nullptr;
    StagedTarget* st = P.sorted + (size_t)b * P.n_cap;
    float* sxr = P.sorted_xr + (size_t)b * P.n_cap;
    auto cnt_inc = [&](int c) -> int { return (int)((atomicAdd(&cnt2[c >> 1], 1u << (16 * (c & 1))) >> (16 * (c & 1))) & 0xffffu); };   // the count before
    for (int i = tid; i < 2048; i += 256) cnt2[i] = 0;
    wg_barrier();
    // (passes 1 and 2 take four targets per thread and trip and load them together: a trip per target had been a memory round trip per target)
    auto cell_xy = [&](float x, float y, bool occ, int& cx, int& cy) -> bool {
        cx = floor_d((double)__fsub_rn(x, P.grid_min_x) * P.inv_cell_w);
        cy = floor_d((double)__fsub_rn(y, P.grid_min_y) * P.inv_cell_h);
        return cx >= 0 && cx < P.grid_cols && cy >= 0 && cy < P.grid_rows && !occ;
    };
    for (int t0 = tid; t0 < n; t0 += 4 * 256) {   // pass 1: cell sizes
        float kx[4], ky[4]; bool oc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int t = min(t0 + 256 * u, n - 1); kx[u] = kps[t].x; ky[u] = kps[t].y; oc[u] = t_occ && t_occ[t]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int cx, cy;
            if (t0 + 256 * u < n && cell_xy(kx[u], ky[u], oc[u], cx, cy)) cnt_inc(cx * P.grid_rows + cy);
        }
    }
    wg_barrier();
    {   // exclusive scan over the cells: 16 cells per thread, a shuffle scan inside the wave, four wave totals through LDS
        const int lane = tid & 63, wv = tid >> 6;
        int c16[16], sum = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const uint32_t w = cnt2[tid * 8 + i]; c16[2 * i] = (int)(w & 0xffffu); c16[2 * i + 1] = (int)(w >> 16); sum += c16[2 * i] + c16[2 * i + 1]; }
        int inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int up = __shfl_up(inc, o); if (lane >= o) inc += up; }
        if (lane == 63) part[wv] = inc;
        wg_barrier();
        int base = inc - sum;
        for (int k = 0; k < wv; ++k) base += part[k];
#pragma unroll
        for (int i = 0; i < 16; ++i) { start[tid * 16 + i] = (uint16_t)base; base += c16[i]; }
        if (tid == 255) { start[4096] = (uint16_t)base; start[4097] = (uint16_t)base; }
        wg_barrier();
        for (int i = tid; i < 2048; i += 256) cnt2[i] = 0;
        wg_barrier();
    }
    for (int t0 = tid; t0 < n; t0 += 4 * 256) {   // pass 2: unordered placement inside the cell
        float kx[4], ky[4]; bool oc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int t = min(t0 + 256 * u, n - 1); kx[u] = kps[t].x; ky[u] = kps[t].y; oc[u] = t_occ && t_occ[t]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + 256 * u;
            int cx, cy;
            if (t >= n || !cell_xy(kx[u], ky[u], oc[u], cx, cy)) continue;
            const int cell = cx * P.grid_rows + cy;
            tmp_t[start[cell] + cnt_inc(cell)] = (uint16_t)t;
        }
    }
    wg_barrier();
    const int used = start[4096];
    for (int p0 = tid; p0 < used; p0 += 4 * 256) {   // pass 3: rank inside the cell by index, final record (four targets per trip, their gathers together)
        int tt[4]; float kx[4], ky[4], xr[4] = {0.f, 0.f, 0.f, 0.f}; int ko[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            tt[u] = tmp_t[min(p0 + 256 * u, used - 1)];
            kx[u] = kps[tt[u]].x; ky[u] = kps[tt[u]].y; ko[u] = kps[tt[u]].octave;
            if (t_xr) xr[u] = t_xr[tt[u]];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (p0 + 256 * u >= used) continue;
            const int t = tt[u];
            int cx, cy;
            cell_xy(kx[u], ky[u], false, cx, cy);
            const int cell = cx * P.grid_rows + cy, s0 = start[cell], s1 = start[cell + 1];
            int rank = 0;
            for (int j = s0; j < s1; ++j) rank += (int)tmp_t[j] < t;
            StagedTarget r;
            r.x = kx[u]; r.y = ky[u]; r.packed = ((uint32_t)ko[u] & 0xffu) | ((uint32_t)cx << 8) | ((uint32_t)cy << 16); r.t = (uint32_t)t;
            st[s0 + rank] = r;
            if (t_xr) sxr[s0 + rank] = xr[u];
        }
    }
    for (int i = tid; i <= ncell; i += 256) P.cell_start[(size_t)b * kCellStride + i] = start[min(i, 4096)];
}

#ifndef PLP_MATCH_LANE_CAND      // candidates a lane collects before it fetches their descriptors = descriptor fetches in flight per lane.  6: 97 VGPRs, one wave
#define PLP_MATCH_LANE_CAND 1    // per SIMD beside two region growers; 3: 72, two; 1: 54, three -- and alone the kernel is FASTER with one (the four matcher calls
#endif                           // 3.84 -> 3.52 ms): its occupancy, not its memory-level parallelism per lane, is what it runs on.  profiles/r03_scheduling_experiments.md
#if PLP_MATCH_LANE_CAND <= 4
#define PLP_TOPK_CELLS_BOUNDS __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8)))
#else
#define PLP_TOPK_CELLS_BOUNDS __launch_bounds__(256)
#endif
constexpr int kLaneCand = PLP_MATCH_LANE_CAND;   // candidate positions a lane collects before it fetches their descriptors

// 16-lane (DPP row) reductions: four queries share a wave
__device__ __forceinline__ uint32_t row16_min_u32(uint32_t v) {
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xf, 0xf, false));   // row_ror:8
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x124, 0xf, 0xf, false));   // row_ror:4
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x122, 0xf, 0xf, false));   // row_ror:2
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x121, 0xf, 0xf, false));   // row_ror:1
    return v;
}
__device__ __forceinline__ int row16_sum_i32(int v) {
    v += __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(v, v, 0x124, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(v, v, 0x122, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(v, v, 0x121, 0xf, 0xf, false);
    return v;
}

// Main path of the point projection matchers.  The workgroup copies the frame's sorted targets (16 B each), their
// stereo coordinates and the cell index into LDS once and reuses them for kQueriesPerBlock queries.  A query is served
// by ONE lane, which walks the grid columns of its window (each column one contiguous range of the sorted array, so only the
// window's cells are touched) and keeps its 8 best keys (distance << 16 | position in visiting order) in registers.  A window at
// the low pyramid levels is 4-5 columns with about five candidates: the earlier 16-lanes-per-query version left two thirds of
// its lanes idle and spent more instructions merging the lanes' lists (8 DPP row minima per query) than finding candidates.
// Consecutive queries sit on the same pyramid level (key points are stored by level), so the lanes of a wave run similar trip counts.
// grid = (ceil(m_cap / kQueriesPerBlock), B), block = 256, dynamic LDS = n_cap * 20 + 2 * kCellStride bytes.
__global__ PLP_TOPK_CELLS_BOUNDS void k_match_topk_cells(MatchProblem P, int qpb) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    unsigned uqb, ub;
    xcd_frame_major(uqb, ub);   // the query blocks of a frame all stage the same sorted target array
    const int tid = threadIdx.x, b = (int)ub;
    const int m = P.q_counts ? min(P.q_counts[b], P.m_cap) : P.m_cap;
    const int q_begin = (int)uqb * qpb;
    if (q_begin >= m) return;
    const int ncell = P.grid_cols * P.grid_rows, rows = P.grid_rows;
    const uint8_t* t_desc = P.t_desc + (size_t)b * P.n_cap * 32;
    // LDS (12 or 16 bytes per target: beside region growing's 85 KB per CU a 50 KB workgroup fitted only once per CU):
    //   sxy[nb] position, sto[nb] = index | octave << 16, sxr[nb] stereo coordinate (only if given), cell index
    const bool has_xr = P.t_x_right != nullptr;
    // The first nb sorted targets of the frame live in LDS (nb = the caller's expected bound of the frame's target count, plp_match_args.
    // t_count_hint, at most the array capacity n_cap): sized by the capacity -- 2064 slots for frames of ~1000 key points -- the workgroup held
    // 33 KB and two of them fitted beside the region growers.  A frame with more targets than nb reads the rest from the sorted array in memory.
    const int nb = P.lds_targets;
    float2* sxy = reinterpret_cast<float2*>(smem);
    uint32_t* sto = reinterpret_cast<uint32_t*>(smem + (size_t)nb * 8);
    float* sxr = reinterpret_cast<float*>(smem + (size_t)nb * 12);
    uint16_t* cs = reinterpret_cast<uint16_t*>(smem + (size_t)nb * (has_xr ? 16 : 12));
    __shared__ uint16_t s_cand[256 * kLaneCand];
    const uint4* g_sorted = reinterpret_cast<const uint4*>(P.sorted + (size_t)b * P.n_cap);   // {x, y, octave | cell, index}
    const float* g_sorted_xr = P.sorted_xr + (size_t)b * P.n_cap;
    {
        const uint16_t* gcs = P.cell_start + (size_t)b * kCellStride;
        const int used = min((int)gcs[ncell], nb);
        for (int i0 = tid; i0 < used; i0 += 4 * 256) {   // four records per thread and trip, loaded together: one memory round trip per 1024 targets, not four
            uint4 r[4]; float xr[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = min(i0 + 256 * u, used - 1); r[u] = g_sorted[i]; if (has_xr) xr[u] = g_sorted_xr[i]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + 256 * u;
                if (i < used) {
                    sxy[i] = make_float2(__uint_as_float(r[u].x), __uint_as_float(r[u].y));
                    sto[i] = (r[u].w & 0xffffu) | ((r[u].z & 0xffu) << 16);
                    if (has_xr) sxr[i] = xr[u];
                }
            }
        }
        const uint32_t* g32 = reinterpret_cast<const uint32_t*>(gcs);
        for (int i = tid; i < (ncell + 2) / 2; i += 256) reinterpret_cast<uint32_t*>(cs)[i] = g32[i];
    }
    wg_barrier();
    // (the staged copies are read through pointers that SAY they are LDS: written as `i < nb ? sxy[i] : ...` the compiler selected between the two POINTERS and
    // loaded through a FLAT instruction -- round 6 keeps FLAT away from LDS in every kernel, profiles/r06_seed_sort.md)
    typedef const __attribute__((address_space(3))) uint32_t* lds_u32; typedef const __attribute__((address_space(3))) float* lds_f32;
    const lds_f32 sxy_l = (lds_f32)sxy; const lds_u32 sto_l = (lds_u32)sto; const lds_f32 sxr_l = (lds_f32)sxr;
    auto t_xy = [&](int i) -> float2 { if (i < nb) return make_float2(sxy_l[2 * i], sxy_l[2 * i + 1]); const uint4 r = g_sorted[i]; return make_float2(__uint_as_float(r.x), __uint_as_float(r.y)); };
    auto t_to = [&](int i) -> uint32_t { if (i < nb) return sto_l[i]; const uint4 r = g_sorted[i]; return (r.w & 0xffffu) | ((r.z & 0xffu) << 16); };   // index | octave << 16
    auto t_xr = [&](int i) -> float { return i < nb ? sxr_l[i] : g_sorted_xr[i]; };
    const uint8_t* q_valid = P.q_valid ? P.q_valid + (size_t)b * P.m_cap : nullptr;
    const int q_end = min(m, q_begin + qpb);
    for (int q = q_begin + tid; q < q_end; q += 256) {
        const bool active = !(q_valid && !q_valid[q]);
        uint32_t top[kMatchK], ovf[kMatchK];   // the 8 best keys, and the next 8: what falls out of (or never reaches) the first list
#pragma unroll
        for (int i = 0; i < kMatchK; ++i) { top[i] = 0xffffffffu; ovf[i] = 0xffffffffu; }
        int passed = 0;
        if (active) {
            const QueryCtx c = make_query(P, q, b);
            if (!c.empty) {
                const uint4* qd = reinterpret_cast<const uint4*>(P.q_desc + ((size_t)b * P.q_desc_stride + q) * 32);
                const uint4 q0 = qd[0], q1 = qd[1];
                const bool check_level = (0 < c.min_level) || (0 <= c.max_level);
                // Two phases so that the descriptor gathers of a lane are in flight together: (1) walk the column
                // ranges and note the positions that pass the geometric tests (up to kLaneCand in an LDS slot of this
                // thread), (2) fetch their descriptors, (3) distances and the best-8 insertion.  A lane that finds more
                // candidates drains its slot and goes on.
                uint16_t* my = s_cand + tid * kLaneCand;
                int nc = 0;
                auto drain = [&]() {
                    uint4 d0[kLaneCand], d1[kLaneCand];
#pragma unroll
                    for (int k = 0; k < kLaneCand; ++k)
                        if (k < nc) {
                            const uint4* d = reinterpret_cast<const uint4*>(t_desc + 32 * (size_t)(t_to(my[k]) & 0xffffu));
                            d0[k] = d[0]; d1[k] = d[1];
                        }
#pragma unroll
                    for (int k = 0; k < kLaneCand; ++k)
                        if (k < nc) {
                            uint32_t key = (hamming256(q0, q1, d0[k], d1[k]) << 16) | (uint32_t)my[k];
                            if (key < top[kMatchK - 1]) {
                                const uint32_t out = top[kMatchK - 1];
                                top[kMatchK - 1] = key;
#pragma unroll
                                for (int e = kMatchK - 1; e > 0; --e) { const uint32_t lo = min(top[e], top[e - 1]), hi = max(top[e], top[e - 1]); top[e - 1] = lo; top[e] = hi; }
                                key = out;
                            }
                            if (key < ovf[kMatchK - 1]) {
                                ovf[kMatchK - 1] = key;
#pragma unroll
                                for (int e = kMatchK - 1; e > 0; --e) { const uint32_t lo = min(ovf[e], ovf[e - 1]), hi = max(ovf[e], ovf[e - 1]); ovf[e - 1] = lo; ovf[e] = hi; }
                            }
                        }
                    passed += nc;
                    nc = 0;
                };
                for (int col = c.min_cx; col <= c.max_cx; ++col) {
                    const int i0 = cs[col * rows + c.min_cy], i1 = cs[col * rows + c.max_cy + 1];
                    for (int i = i0; i < i1; ++i) {
                        const float2 sp = t_xy(i);
                        if (check_level) {
                            const int oct = (int)(t_to(i) >> 16);
                            if (oct < c.min_level) continue;
                            if (0 <= c.max_level && c.max_level < oct) continue;
                        }
                        if (!(fabsf(__fsub_rn(sp.x, c.rx)) < c.mg && fabsf(__fsub_rn(sp.y, c.ry)) < c.mg)) continue;
                        if (has_xr) {
                            const float xr = t_xr(i);
                            if (0 < xr && c.mg < fabsf(__fsub_rn(c.xr, xr))) continue;
                        }
                        my[nc++] = (uint16_t)i;
                        if (nc == kLaneCand) drain();
                    }
                }
                drain();
            }
        }
        uint32_t e[kMatchK];
#pragma unroll
        for (int k = 0; k < kMatchK; ++k) {
            e[k] = 0xffffffffu;
            if (top[k] != 0xffffffffu) {
                const uint32_t to = t_to((int)(top[k] & 0xffffu));
                e[k] = ((top[k] >> 16) << 20) | (((to >> 16) & 15u) << 16) | (to & 0xffffu);
            }
        }
        static_assert(kMatchK == 8, "two 16-byte stores per query");
        uint4* out = reinterpret_cast<uint4*>(P.klist + ((size_t)b * P.m_cap + q) * kMatchK);
        out[0] = make_uint4(e[0], e[1], e[2], e[3]);
        out[1] = make_uint4(e[4], e[5], e[6], e[7]);
        P.kcount[(size_t)b * P.m_cap + q] = active ? passed : -1;
        if (passed > kMatchK) {   // ranks 9..16 for the resolver: a first list whose entries are all taken rarely needs an exact rescan then
#pragma unroll
            for (int k = 0; k < kMatchK; ++k) {
                e[k] = 0xffffffffu;
                if (ovf[k] != 0xffffffffu) {
                    const uint32_t to = t_to((int)(ovf[k] & 0xffffu));
                    e[k] = ((ovf[k] >> 16) << 20) | (((to >> 16) & 15u) << 16) | (to & 0xffffu);
                }
            }
            uint4* out2 = reinterpret_cast<uint4*>(P.klist2 + ((size_t)b * P.m_cap + q) * kMatchK);
            out2[0] = make_uint4(e[0], e[1], e[2], e[3]);
            out2[1] = make_uint4(e[4], e[5], e[6], e[7]);
        }
    }
}

// Brute-force mode: the frame's 32-byte descriptors are staged in LDS, one wave per query scans them all.
// grid = (ceil(m_cap / kQueriesPerBlock), B), block = 256, dynamic LDS = n_cap * 32 bytes.
__global__ __launch_bounds__(256) void k_match_topk_lds(MatchProblem P) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), b = blockIdx.y;
    const int m = P.q_counts ? min(P.q_counts[b], P.m_cap) : P.m_cap;
    const int q_begin = blockIdx.x * kQueriesPerBlock;
    if (q_begin >= m) return;
    const int n = P.t_counts ? min(P.t_counts[b], P.n_cap) : P.n_cap;
    const uint8_t* t_desc = P.t_desc + (size_t)b * P.n_cap * 32;
    uint4* sdesc = reinterpret_cast<uint4*>(smem);
    {
        const uint4* src = reinterpret_cast<const uint4*>(t_desc);
        for (int i = tid; i < 2 * n; i += 256) sdesc[i] = src[i];
    }
    wg_barrier();
    const uint8_t* q_valid = P.q_valid ? P.q_valid + (size_t)b * P.m_cap : nullptr;
    for (int q = q_begin + wv; q < min(m, q_begin + kQueriesPerBlock); q += 4) {
        uint32_t* klist = P.klist + ((size_t)b * P.m_cap + q) * kMatchK;
        int32_t* kcount = P.kcount + (size_t)b * P.m_cap + q;
        if (q_valid && !q_valid[q]) { if (lane == 0) *kcount = -1; continue; }
        const uint4* qd = reinterpret_cast<const uint4*>(P.q_desc + ((size_t)b * P.q_desc_stride + q) * 32);
        const uint4 q0 = qd[0], q1 = qd[1];
        unsigned long long top[kMatchK];
#pragma unroll
        for (int i = 0; i < kMatchK; ++i) top[i] = ~0ull;
        int passed = 0;
        for (int t = lane; t < n; t += 64) {
            const unsigned dist = hamming256(q0, q1, sdesc[2 * t], sdesc[2 * t + 1]);
            ++passed;
            topk_insert(top, ((unsigned long long)dist << 32) | ((unsigned long long)(unsigned)t << 4));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) passed += __shfl_xor(passed, o);
        topk_merge_store(top, lane, klist);
        if (lane == 0) *kcount = passed;
    }
}

// accept rules of the three matchers; best/second are (distance, octave) of the two best free candidates
template <int FAM = kFamAny>
__device__ __forceinline__ bool accept(const MatchProblem& P, unsigned best, int best_lvl, unsigned second, int second_lvl) {
    const int mode = fam_mode<FAM>(P.mode);
    if (mode == PLP_MATCH_MODE_LANDMARKS || mode == PLP_MATCH_MODE_LANDMARKS_LINE) {
        if (!(best <= 100u)) return false;
        if (best_lvl == second_lvl && (float)best > __fmul_rn(P.lowe_ratio, (float)second)) return false;
        return true;
    }
    if (is_last_frame_mode(mode)) return best <= (P.hamm_dist_thr > 0 ? (unsigned)P.hamm_dist_thr : 100u);
    if (mode == PLP_MATCH_MODE_TRIANGULATION) return true;                 // every gate is part of the candidate test
    if (50u < best) return false;                                            // brute force: HAMMING_DIST_THR_LOW
    if (__fmul_rn(P.lowe_ratio, (float)second) < (float)best) return false;
    return true;
}

// grid = (B), block = 256.  LDS: owner[2][n_cap] (dynamic).
// kSorted: the windowed point modes after k_match_prep (a dry list is rescanned over the window's ranges of the sorted array); the other
// instantiation rescans through candidate_key() and is the only one that carries its registers (all modes' gates, f64 epipolar tests).
template <bool kSorted, int FAM = kFamAny>
__device__ __forceinline__ void match_resolve_body(const MatchProblem& P) {
    const int mode = fam_mode<FAM>(P.mode);
    extern __shared__ int32_t lds[];
    __shared__ int s_changed, s_num, s_hist[32], s_valid_bin[32], s_full_n, s_claim_tmp[256], s_sort_ws[48];
    __shared__ unsigned s_sort_idx[32];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), b = blockIdx.x;
    const int m = P.q_counts ? min(P.q_counts[b], P.m_cap) : P.m_cap;
    const int n = P.t_counts ? min(P.t_counts[b], P.n_cap) : P.n_cap;
    int32_t* owner_final = lds;                 // smallest claimant among the chunks already finished
    int32_t* owner_prev = lds + P.n_cap;        // claims inside the current chunk, previous / current inner iteration
    int32_t* owner_next = lds + 2 * P.n_cap;
    const uint32_t* klist = P.klist + (size_t)b * P.m_cap * kMatchK;
    const int32_t* kcount = P.kcount + (size_t)b * P.m_cap;
    int32_t* claim = P.claim + (size_t)b * P.m_cap;           // per query: claimed target or -1
    const uint8_t* has_obs = P.q_has_obs ? P.q_has_obs + (size_t)b * P.m_cap : nullptr;
    const plp_keypoint* kps = P.t_kps ? P.t_kps + (size_t)b * P.n_cap : nullptr;
    const uint8_t* t_desc = P.t_desc + (size_t)b * P.n_cap * 32;
    const float* t_xr = P.t_x_right ? P.t_x_right + (size_t)b * P.n_cap : nullptr;
    const uint8_t* t_occ = P.t_occupied ? P.t_occupied + (size_t)b * P.n_cap : nullptr;
    int32_t* out = P.out_match + (size_t)b * P.n_cap;

    for (int t = tid; t < n; t += 256) { owner_final[t] = 0x7fffffff; out[t] = -1; }
    for (int q = tid; q < m; q += 256) claim[q] = -1;
    wg_barrier();

    int32_t* full_list = P.full_list + (size_t)b * P.m_cap;   // queries whose truncated best-K list ran dry
    const int need = (is_last_frame_mode(mode) || mode == PLP_MATCH_MODE_TRIANGULATION) ? 1 : 2;   // the last-frame matcher has no second-best test
    const bool blocks_always = !has_obs || mode == PLP_MATCH_MODE_BRUTE_FORCE || is_group_mode(mode);
    const unsigned t_flip = mode == PLP_MATCH_MODE_TRIANGULATION ? 0xffffu : 0u;   // that mode orders equal distances by DESCENDING index
    constexpr bool use_sorted = kSorted;
    const StagedTarget* sorted = P.sorted + (size_t)b * P.n_cap;
    const float* sorted_xr = P.sorted_xr + (size_t)b * P.n_cap;
    const uint16_t* g_cell_start = P.cell_start + (size_t)b * kCellStride;
    // "claim[q] = best free candidate given the claims of the queries before q", 256 queries (one chunk) at a time in
    // index order.  Inside a chunk the claims are iterated to their fixed point (claims of its first r queries are
    // final after r iterations, in practice after 2-3); the chunks before it are final already, so one pass over the
    // chunks gives the sequential answer.  (The first version iterated whole sweeps over all chunks: ~7 sweeps.)
    auto taken = [&](int t, int q, int) -> bool { return owner_final[t] != 0x7fffffff || owner_prev[t] < q; };
    {
        for (int chunk_start = 0; chunk_start < m; chunk_start += 256) {
          const int q = chunk_start + tid;
          int my_claim = -1;
          for (int t = tid; t < n; t += 256) owner_prev[t] = 0x7fffffff;
          // the query's best-K list and candidate count: read once per chunk (two 16-byte loads), the iterations work on registers
          constexpr int kList = kSorted ? 2 * kMatchK : kMatchK;   // k_match_topk_cells also leaves ranks 9..16 of a crowded window
          uint32_t e8[kList];
          int cnt = 0;
          if (q < m) {
              cnt = kcount[q];
              const uint4* kp = reinterpret_cast<const uint4*>(klist + (size_t)q * kMatchK);
              const uint4 lo = kp[0], hi = kp[1];
              e8[0] = lo.x; e8[1] = lo.y; e8[2] = lo.z; e8[3] = lo.w; e8[4] = hi.x; e8[5] = hi.y; e8[6] = hi.z; e8[7] = hi.w;
              if constexpr (kSorted) {
#pragma unroll
                  for (int k = kMatchK; k < kList; ++k) e8[k] = 0xffffffffu;
                  if (cnt > kMatchK) {
                      const uint4* kp2 = reinterpret_cast<const uint4*>(P.klist2 + ((size_t)b * P.m_cap + q) * kMatchK);
                      const uint4 lo2 = kp2[0], hi2 = kp2[1];
                      e8[8] = lo2.x; e8[9] = lo2.y; e8[10] = lo2.z; e8[11] = lo2.w; e8[12] = hi2.x; e8[13] = hi2.y; e8[14] = hi2.z; e8[15] = hi2.w;
                  }
              }
          }
          for (int inner = 0; inner <= 256; ++inner) {
            if (tid == 0) { s_full_n = 0; s_changed = 0; }
            for (int t = tid; t < n; t += 256) owner_next[t] = 0x7fffffff;
            wg_barrier();
            int new_claim = -1;
            bool decided = false;
            if (q < m) {
                if (cnt <= 0) decided = true;
                else {
                    const int have = min(cnt, kList);
                    unsigned best = 256, second = 256;
                    int best_lvl = -1, second_lvl = -1, best_t = -1, found = 0;
#pragma unroll
                    for (int e = 0; e < kList; ++e) {
                        if (e >= have || found >= need) continue;
                        const int t = (int)((e8[e] & 0xffff) ^ t_flip);
                        if (taken(t, q, chunk_start)) continue;
                        if (found == 0) { best = e8[e] >> 20; best_lvl = (int)((e8[e] >> 16) & 15); best_t = t; }
                        else { second = e8[e] >> 20; second_lvl = (int)((e8[e] >> 16) & 15); }
                        ++found;
                    }
                    // The truncated list ran dry (every entry taken, or only the best found) although the window holds more
                    // candidates.  Whatever lies beyond the list is at least as far as its last entry, d8 -- often that alone
                    // decides: no acceptable best exists beyond it (d8 above the mode's distance threshold), or the best that was
                    // found passes (or fails) its test against ANY second-best >= d8.  Only the rest needs the exact rescan.
                    bool rescan = found < need && cnt > kList;
                    if (rescan && mode != PLP_MATCH_MODE_TRIANGULATION) {
                        const unsigned d8 = e8[kList - 1] >> 20;   // the distance of the list's last entry: nothing beyond it is closer
                        const bool lm = mode == PLP_MATCH_MODE_LANDMARKS || mode == PLP_MATCH_MODE_LANDMARKS_LINE;
                        const unsigned thr = (lm || is_last_frame_mode(mode)) ? (is_last_frame_mode(mode) && P.hamm_dist_thr > 0 ? (unsigned)P.hamm_dist_thr : 100u) : 50u;
                        if (found == 0) { if (d8 > thr) rescan = false; }                       // nothing acceptable is left: no claim
                        else if (best > thr) rescan = false;                                      // (found == 1, need == 2) rejected whatever the second is
                        else if (lm) { if (!((float)best > __fmul_rn(P.lowe_ratio, (float)d8))) { rescan = false; second = d8; second_lvl = -2; } }
                        else if (!(__fmul_rn(P.lowe_ratio, (float)d8) < (float)best)) { rescan = false; second = d8; }
                    }
                    if (rescan) full_list[atomicAdd(&s_full_n, 1)] = q;   // exact rescan below
                    else { decided = true; if (found > 0 && accept<FAM>(P, best, best_lvl, second, second_lvl)) new_claim = best_t; }
                }
            }
            wg_barrier();
            // rare: exact two-best over the query's whole window with the occupancy filter, one wave per query
            const int nf = s_full_n;
            if (tid == 0 && P.dbg && nf) atomicAdd(&P.dbg[0], nf);
            for (int f = wv; f < nf; f += 4) {
                const int fq = full_list[f];
                const QueryCtx c = make_query<FAM>(P, fq, b);
                const uint4* qd = reinterpret_cast<const uint4*>(P.q_desc + ((size_t)b * P.q_desc_stride + fq) * 32);
                const uint4 q0 = qd[0], q1 = qd[1];
                unsigned long long k0 = ~0ull, k1 = ~0ull;
                if constexpr (use_sorted) {
                    if (!c.empty) {
                        const bool check_level = (0 < c.min_level) || (0 <= c.max_level);
                        for (int col = c.min_cx; col <= c.max_cx; ++col) {
                            const int i0 = g_cell_start[col * P.grid_rows + c.min_cy], i1 = g_cell_start[col * P.grid_rows + c.max_cy + 1];
                            for (int i = i0 + lane; i < i1; i += 64) {
                                const StagedTarget s = sorted[i];
                                const int oct = (int)(s.packed & 0xff);
                                if (check_level) {
                                    if (oct < c.min_level) continue;
                                    if (0 <= c.max_level && c.max_level < oct) continue;
                                }
                                if (!(fabsf(__fsub_rn(s.x, c.rx)) < c.mg && fabsf(__fsub_rn(s.y, c.ry)) < c.mg)) continue;
                                if (t_xr) {
                                    const float xr = sorted_xr[i];
                                    if (0 < xr && c.mg < fabsf(__fsub_rn(c.xr, xr))) continue;
                                }
                                if (taken((int)s.t, fq, chunk_start)) continue;
                                const uint4* d = reinterpret_cast<const uint4*>(t_desc + 32 * (size_t)s.t);
                                const unsigned dist = hamming256(q0, q1, d[0], d[1]);
                                // visiting order = position in the (column, row, index)-sorted array
                                const unsigned long long key = ((unsigned long long)dist << 32) | ((unsigned long long)(unsigned)i << 4) | (unsigned)(oct & 15);
                                if (key < k0) { k1 = k0; k0 = key; } else if (key < k1) k1 = key;
                            }
                        }
                    }
                } else if (!(c.windowed && c.empty)) {
                    for (int t = lane; t < n; t += 64) {
                        if (taken(t, fq, chunk_start)) continue;
                        const unsigned long long key = candidate_key<FAM>(P, c, t, kps, t_desc, t_xr, t_occ, q0, q1);
                        if (key < k0) { k1 = k0; k0 = key; } else if (key < k1) k1 = key;
                    }
                }
                const unsigned long long g0 = wave_min_u64(k0);
                const unsigned long long g1 = wave_min_u64(k0 == g0 ? k1 : k0);
                if (lane == 0) {
                    int nc = -1;
                    if (g0 != ~0ull) {
                        const unsigned second = g1 != ~0ull ? (unsigned)(g1 >> 32) : 256u;
                        const int second_lvl = g1 != ~0ull ? (int)(g1 & 15) : -1;
                        if (accept<FAM>(P, (unsigned)(g0 >> 32), (int)(g0 & 15), second, second_lvl))
                            nc = use_sorted ? (int)sorted[(g0 >> 4) & 0xffff].t : (int)(((g0 >> 4) & 0xffff) ^ t_flip);
                    }
                    s_claim_tmp[fq - chunk_start] = nc;
                }
            }
            wg_barrier();
            if (q < m) {
                if (!decided) new_claim = s_claim_tmp[tid];
                if (my_claim != new_claim) { my_claim = new_claim; s_changed = 1; }
                if (new_claim >= 0 && (blocks_always || has_obs[q])) atomicMin(&owner_next[new_claim], q);
            }
            wg_barrier();
            { int32_t* t = owner_prev; owner_prev = owner_next; owner_next = t; }
            const int changed = s_changed;
            wg_barrier();
            if (!changed) break;
          }
          if (tid == 0 && P.dbg) atomicAdd(&P.dbg[1], 1);
          // the chunk is final: publish its claims
          if (q < m) {
              claim[q] = my_claim;
              if (my_claim >= 0 && (blocks_always || has_obs[q])) atomicMin(&owner_final[my_claim], q);
          }
          wg_barrier();
        }
    }

    // ---- results: last writer per key point, number of accepted queries, delta-angle histogram check
    if (tid == 0) s_num = 0;
    for (int i = tid; i < 32; i += 256) { s_hist[i] = 0; s_valid_bin[i] = 0; }
    wg_barrier();
    const bool angle_check = P.check_orientation && (mode == PLP_MATCH_MODE_LAST_FRAME || mode == PLP_MATCH_MODE_BRUTE_FORCE || is_group_mode(mode));
    const float* q_angle = P.q_angle ? P.q_angle + (size_t)b * P.m_cap : nullptr;
    const float* t_angle = P.t_angle ? P.t_angle + (size_t)b * P.n_cap : nullptr;
    auto bin_of = [&](int q, int t) -> int {
        const float ta = kps ? kps[t].angle : t_angle[t];
        float delta = (mode == PLP_MATCH_MODE_LAST_FRAME || is_group_mode(mode)) ? __fsub_rn(q_angle[q], ta) : __fsub_rn(ta, q_angle[q]);
        if (delta < 0.0) delta = (float)((double)delta + 360.0);
        if (360.0 <= delta) delta = (float)((double)delta - 360.0);
        // angles outside [0, 360) can leave delta negative: the reference throws there (angle_histogram_.at(bin)); here such a
        // match lands in bin 31, which is never among the kept bins
        const int bin = __float2int_rn(__fmul_rn(delta, 1.0f / 30));
        return (unsigned)bin > 31u ? 31 : bin;
    };
    int my = 0;
    for (int q = tid; q < m; q += 256) {
        const int t = claim[q];
        if (t < 0) continue;
        ++my;
        atomicMax(&out[t], q);
        if (angle_check) atomicAdd(&s_hist[min(bin_of(q, t), 31)], 1);
    }
    if (my) atomicAdd(&s_num, my);
    wg_barrier();
    if (angle_check) {
        if (tid == 0) {   // the first 3 of the 30 bins as the reference's std::sort by size orders them (angle_checker.h:165-176), ties included
            libstdcxx::index_sort_by_size(s_hist, 30, s_sort_idx, s_sort_ws);
            for (int r = 0; r < 3; ++r) s_valid_bin[s_sort_idx[r]] = 1;
        }
        wg_barrier();
        int bad = 0;
        for (int q = tid; q < m; q += 256) {
            const int t = claim[q];
            if (t < 0) continue;
            if (!s_valid_bin[min(bin_of(q, t), 31)]) { out[t] = (P.flags & PLP_MATCH_FLAG_MARK_INVALIDATED) ? -2 : -1; ++bad; }
        }
        if (bad) atomicSub(&s_num, bad);
        wg_barrier();
    }
    if (tid == 0) P.out_num[b] = s_num;
}
// kernels around the one body: the sorted instantiation is the hot one (point matchers of every frame) and is held to 128 registers (4 waves per
// SIMD beside its 24 KB of LDS).  Held to 96 (ranks 9..16 of a query's list fetched on demand instead of kept in registers) it runs two waves per
// SIMD beside two region growers instead of one -- 1.3 instead of 4.1 ms per call IN the step -- but 0.5 instead of 0.36 ms alone, and the step
// did not move (the matcher stream is not its critical path): not adopted, profiles/r03_scheduling_experiments.md.  The generic ones carry every
// mode's gates and take the registers they need instead of spilling
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_match_resolve_sorted(MatchProblem P) { match_resolve_body<true, kFamGrid>(P); }
template <int FAM>
__global__ __launch_bounds__(256) void k_match_resolve_generic(MatchProblem P) { match_resolve_body<false, FAM>(P); }
// (A 64-thread instantiation for the key-line matchers -- one wave per problem -- runs 0.1 instead of 2.2 ms per
// call INSIDE the step, where the 256-thread workgroups get one slot per CU, but 0.21 instead of 0.075 ms alone, and the step follows the isolated
// times: measured, not kept; profiles/r03_scheduling_experiments.md.)

// ------------------------------------------------------------------------------------------
// K16  full Hamming matrix: dist[q][t] (u16), 64 x 64 tile per workgroup, descriptors staged in LDS.
// grid = (ceil(nt/64), ceil(nq/64)), block = 256.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hamming_matrix(const uint8_t* __restrict__ qd, int nq, const uint8_t* __restrict__ td, int nt,
                                                        uint16_t* __restrict__ dist) {
    __shared__ uint32_t sq[64 * 9], st[64 * 9];   // 8 dwords per descriptor, +1 pad against bank conflicts
    const int tid = threadIdx.x;
    const int q0 = blockIdx.y * 64, t0 = blockIdx.x * 64;
    for (int i = tid; i < 64 * 8; i += 256) {
        const int r = i >> 3, w = i & 7;
        sq[r * 9 + w] = q0 + r < nq ? reinterpret_cast<const uint32_t*>(qd)[(size_t)(q0 + r) * 8 + w] : 0u;
        st[r * 9 + w] = t0 + r < nt ? reinterpret_cast<const uint32_t*>(td)[(size_t)(t0 + r) * 8 + w] : 0u;
    }
    wg_barrier();
    const int tx = tid & 63, ty = __builtin_amdgcn_readfirstlane(tid >> 6);   // thread: target tx, queries ty, ty+4, ... (ty is the wave's index: scalar)
    uint32_t tv[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) tv[w] = st[tx * 9 + w];
    for (int r = ty; r < 64; r += 4) {
        unsigned d = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) d += __popc(sq[r * 9 + w] ^ tv[w]);
        if (q0 + r < nq && t0 + tx < nt) dist[(size_t)(q0 + r) * nt + t0 + tx] = (uint16_t)d;
    }
}

// ------------------------------------------------------------------------------------------
// fuse::replace_duplication, search part (fuse.cc:169-298): independent queries, one wave each.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_match_fuse(MatchProblem P) {
    const int lane = threadIdx.x & 63, b = blockIdx.y;
    const int q = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int m = P.q_counts ? min(P.q_counts[b], P.m_cap) : P.m_cap;
    if (q >= m) return;
    int32_t* out = P.out_query_best + (size_t)b * P.m_cap + q;
    const uint8_t* q_valid = P.q_valid ? P.q_valid + (size_t)b * P.m_cap : nullptr;
    if (q_valid && !q_valid[q]) { if (lane == 0) *out = -1; return; }
    const int n = P.t_counts ? min(P.t_counts[b], P.n_cap) : P.n_cap;
    const plp_keypoint* kps = P.t_kps + (size_t)b * P.n_cap;
    const uint8_t* t_desc = P.t_desc + (size_t)b * P.n_cap * 32;
    const float* t_xr = P.t_x_right ? P.t_x_right + (size_t)b * P.n_cap : nullptr;
    const QueryCtx c = make_query(P, q, b);
    const uint4* qd = reinterpret_cast<const uint4*>(P.q_desc + ((size_t)b * P.q_desc_stride + q) * 32);
    const uint4 q0 = qd[0], q1 = qd[1];
    unsigned long long best = ~0ull;
    if (!c.empty)
        for (int t = lane; t < n; t += 64) {
            const unsigned long long key = candidate_key(P, c, t, kps, t_desc, t_xr, nullptr, q0, q1);
            best = key < best ? key : best;
        }
    best = wave_min_u64(best);
    const unsigned thr = P.hamm_dist_thr > 0 ? (unsigned)P.hamm_dist_thr : 50u;
    if (lane == 0) *out = (best != ~0ull && (unsigned)(best >> 32) <= thr) ? (int32_t)((best >> 4) & 0xffff) : -1;
}

// ------------------------------------------------------------------------------------------
// area::match_in_consistent_area (area.cc:33-153): the "steal if strictly closer" bookkeeping makes every query depend
// on the distances recorded by all earlier ones, and the matcher only runs during monocular initialisation, so it is
// replayed literally: one wave per problem walks frame 1 in order, its 64 lanes scan frame 2.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_match_area(AreaArgs A) {
    __shared__ int hist[32], valid_bin[32], sort_ws[48];
    __shared__ unsigned sort_idx[32];
    const int lane = threadIdx.x;
    uint32_t* mdist = A.scratch;            // matched_dists_in_frm_2
    int32_t* m1in2 = reinterpret_cast<int32_t*>(A.scratch + A.n2);   // matched_indices_1_in_frm_2
    for (int i = lane; i < A.n2; i += 64) { mdist[i] = 256u; m1in2[i] = -1; }
    for (int i = lane; i < A.n1; i += 64) A.matched_2_in_1[i] = -1;
    if (lane < 32) { hist[lane] = 0; valid_bin[lane] = 0; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    auto bin_of = [&](int i1, int i2) -> int {
        float delta = __fsub_rn(A.kps1[i1].angle, A.kps2[i2].angle);
        if (delta < 0.0) delta = (float)((double)delta + 360.0);
        if (360.0 <= delta) delta = (float)((double)delta - 360.0);
        const int bin = __float2int_rn(__fmul_rn(delta, 1.0f / 30));
        return (unsigned)bin > 31u ? 31 : bin;
    };
    for (int i1 = 0; i1 < A.n1; ++i1) {
        const plp_keypoint k1 = A.kps1[i1];
        if (0 < k1.octave) continue;
        const float rx = A.prev_pts[2 * i1], ry = A.prev_pts[2 * i1 + 1], mg = A.margin;
        const int min_cx = max(0, floor_d((double)__fsub_rn(__fsub_rn(rx, A.grid_min_x), mg) * A.inv_cell_w));
        const int max_cx = min(A.grid_cols - 1, ceil_d((double)__fadd_rn(__fsub_rn(rx, A.grid_min_x), mg) * A.inv_cell_w));
        const int min_cy = max(0, floor_d((double)__fsub_rn(__fsub_rn(ry, A.grid_min_y), mg) * A.inv_cell_h));
        const int max_cy = min(A.grid_rows - 1, ceil_d((double)__fadd_rn(__fsub_rn(ry, A.grid_min_y), mg) * A.inv_cell_h));
        if (A.grid_cols <= min_cx || max_cx < 0 || A.grid_rows <= min_cy || max_cy < 0) continue;
        const uint4* qd = reinterpret_cast<const uint4*>(A.desc1 + 32 * (size_t)i1);
        const uint4 q0 = qd[0], q1 = qd[1];
        unsigned long long k0 = ~0ull, k1k = ~0ull;
        for (int t = lane; t < A.n2; t += 64) {
            const plp_keypoint k = A.kps2[t];
            const int cx = floor_d((double)__fsub_rn(k.x, A.grid_min_x) * A.inv_cell_w), cy = floor_d((double)__fsub_rn(k.y, A.grid_min_y) * A.inv_cell_h);
            if (cx < 0 || cx >= A.grid_cols || cy < 0 || cy >= A.grid_rows) continue;
            if (cx < min_cx || cx > max_cx || cy < min_cy || cy > max_cy) continue;
            if (k.octave < 0 || 0 < k.octave) continue;                      // min_level = max_level = 0
            if (!(fabsf(__fsub_rn(k.x, rx)) < mg && fabsf(__fsub_rn(k.y, ry)) < mg)) continue;
            const uint4* d = reinterpret_cast<const uint4*>(A.desc2 + 32 * (size_t)t);
            const unsigned dist = hamming256(q0, q1, d[0], d[1]);
            if (mdist[t] <= dist) continue;                                  // already matched at least as closely (:72)
            const unsigned long long key = ((unsigned long long)dist << 32) | ((unsigned long long)(((unsigned)(cx * A.grid_rows + cy) << 16) | (unsigned)t));
            if (key < k0) { k1k = k0; k0 = key; } else if (key < k1k) k1k = key;
        }
        const unsigned long long g0 = wave_min_u64(k0);
        const unsigned long long g1 = wave_min_u64(k0 == g0 ? k1k : k0);
        if (g0 == ~0ull) continue;
        const unsigned best = (unsigned)(g0 >> 32), second = g1 != ~0ull ? (unsigned)(g1 >> 32) : 256u;
        if (50u < best) continue;
        if (__fmul_rn((float)second, A.lowe_ratio) < (float)best) continue;
        const int best_i2 = (int)(g0 & 0xffff);
        if (lane == 0) {
            const int prev_i1 = m1in2[best_i2];
            if (0 <= prev_i1) A.matched_2_in_1[prev_i1] = -1;
            A.matched_2_in_1[i1] = best_i2;
            m1in2[best_i2] = i1;
            mdist[best_i2] = best;
            if (A.check_orientation) hist[bin_of(i1, best_i2)] += 1;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    // num_matches = matches still standing; the orientation check removes the ones outside the 3 fullest bins
    if (A.check_orientation && lane == 0) {
        libstdcxx::index_sort_by_size(hist, 30, sort_idx, sort_ws);   // the reference's std::sort of the bins by size, ties included
        for (int r = 0; r < 3; ++r) valid_bin[sort_idx[r]] = 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    int cnt = 0;
    for (int i1 = lane; i1 < A.n1; i1 += 64) {
        const int i2 = A.matched_2_in_1[i1];
        if (i2 < 0) continue;
        if (A.check_orientation && !valid_bin[bin_of(i1, i2)]) { A.matched_2_in_1[i1] = -1; continue; }
        ++cnt;
        A.prev_pts[2 * i1] = A.kps2[i2].x; A.prev_pts[2 * i1 + 1] = A.kps2[i2].y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0) *A.num_matches = cnt;
}

void launch_match_area(hipStream_t st, const AreaArgs& A) { hipLaunchKernelGGL(k_match_area, dim3(1), dim3(64), 0, st, A); }

static bool is_line_mode_host(int mode) { return mode == PLP_MATCH_MODE_LANDMARKS_LINE || mode == PLP_MATCH_MODE_LAST_FRAME_LINE || mode == PLP_MATCH_MODE_FUSE_LINE; }

void launch_match(hipStream_t st, const MatchProblem& P, int B) {
    if (P.mode == PLP_MATCH_MODE_FUSE || P.mode == PLP_MATCH_MODE_FUSE_LINE) { hipLaunchKernelGGL(k_match_fuse, dim3((P.m_cap + 3) / 4, B), dim3(256), 0, st, P); return; }
    MatchProblem Q = P;
    Q.sorted_valid = 0;
    const bool windowed = P.mode == PLP_MATCH_MODE_LANDMARKS || P.mode == PLP_MATCH_MODE_LAST_FRAME;
    const bool line = is_line_mode_host(P.mode) || P.mode == PLP_MATCH_MODE_BOW || P.mode == PLP_MATCH_MODE_TRIANGULATION;
    const int fam = is_line_mode_host(P.mode) ? kFamLine : (P.mode == PLP_MATCH_MODE_BOW || P.mode == PLP_MATCH_MODE_TRIANGULATION) ? kFamGroup : kFamPoint;
    const size_t staged = windowed ? (size_t)P.lds_targets * (P.t_x_right ? 16 : 12) + 2 * kCellStride : (size_t)P.n_cap * 32;
    const dim3 qgrid((P.m_cap + kQueriesPerBlock - 1) / kQueriesPerBlock, B);
    if (!line && windowed && staged <= 64 * 1024 && P.grid_cols <= 255 && P.grid_rows <= 255) {
        hipLaunchKernelGGL(k_match_prep, dim3(B), dim3(256), 0, st, P);
        Q.sorted_valid = 1;
        // queries per workgroup = queries that share one staging of the frame's targets.  Batches: 512 (alone the kernel is 3 % faster with 256 -- more
        // workgroups in flight -- but the step is 0.7 % faster with 512, six passes each: half as many stagings beside the region growers).  A single
        // frame or a few (the synchronous host-pointer entry): the CHIP is empty, so many small workgroups (PLP_MATCH_QPB overrides both).
        static const int qpb_env = [] { const char* e = getenv("PLP_MATCH_QPB"); const int v = e ? atoi(e) : 0; return v >= 16 && v % 16 == 0 ? v : 0; }();
        const int qpb = qpb_env ? qpb_env : (B >= 64 ? 512 : 32);   // one frame: last-frame matcher 0.40 ms with 32, 0.43 with 64, 0.46 with 256, 0.59 with 512
        hipLaunchKernelGGL(k_match_topk_cells, dim3((P.m_cap + qpb - 1) / qpb, B), dim3(256), staged, st, P, qpb);
    } else if (!line && !windowed && staged <= 64 * 1024) {
        hipLaunchKernelGGL(k_match_topk_lds, qgrid, dim3(256), staged, st, P);
    } else {
        const int gx_full = (P.m_cap + 3) / 4, gx_min = std::max(16, (8192 + B - 1) / B);   // keep >= ~8K workgroups in flight
        if (P.n_cap <= 512 && B >= 64) {   // small target sets, many frames
            const dim3 g(std::min((P.m_cap + 63) / 64, 2), B);
            if (fam == kFamLine) hipLaunchKernelGGL(k_match_topk_lanes<kFamLine>, g, dim3(64), 0, st, P);
            else if (fam == kFamGroup) hipLaunchKernelGGL(k_match_topk_lanes<kFamGroup>, g, dim3(64), 0, st, P);
            else hipLaunchKernelGGL(k_match_topk_lanes<kFamPoint>, g, dim3(64), 0, st, P);
        } else {
            const dim3 g(std::min(gx_full, gx_min), B);
            if (fam == kFamLine) hipLaunchKernelGGL(k_match_topk<kFamLine>, g, dim3(256), 0, st, P);
            else if (fam == kFamGroup) hipLaunchKernelGGL(k_match_topk<kFamGroup>, g, dim3(256), 0, st, P);
            else hipLaunchKernelGGL(k_match_topk<kFamPoint>, g, dim3(256), 0, st, P);
        }
    }
    const size_t owners = (size_t)P.n_cap * 12;
    if (Q.sorted_valid) hipLaunchKernelGGL(k_match_resolve_sorted, dim3(B), dim3(256), owners, st, Q);
    else if (fam == kFamLine) hipLaunchKernelGGL(k_match_resolve_generic<kFamLine>, dim3(B), dim3(256), owners, st, Q);
    else if (fam == kFamGroup) hipLaunchKernelGGL(k_match_resolve_generic<kFamGroup>, dim3(B), dim3(256), owners, st, Q);
    else hipLaunchKernelGGL(k_match_resolve_generic<kFamPoint>, dim3(B), dim3(256), owners, st, Q);
}

// up to 8192 targets: 96 KB of owner arrays.  The attribute belongs to the function ON THE CURRENT DEVICE: called by
// plp_matcher_create after hipSetDevice, once per context.
hipError_t configure_match_kernels() {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_match_resolve_sorted), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 12);
    if (e != hipSuccess) return e;
    for (const void* f : {reinterpret_cast<const void*>(k_match_resolve_generic<kFamLine>), reinterpret_cast<const void*>(k_match_resolve_generic<kFamGroup>),
                          reinterpret_cast<const void*>(k_match_resolve_generic<kFamPoint>)}) {
        const hipError_t e2 = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 12);
        if (e2 != hipSuccess) return e2;
    }
    return hipSuccess;
}

void launch_hamming_matrix(hipStream_t st, const uint8_t* q, int nq, const uint8_t* t, int nt, uint16_t* dist) {
    hipLaunchKernelGGL(k_hamming_matrix, dim3((nt + 63) / 64, (nq + 63) / 64), dim3(256), 0, st, q, nq, t, nt, dist);
}

// ------------------------------------------------------------------------------------------
// BinaryDescriptorMatcher::match — exact 1-NN over LBD descriptors (reference
// src/PLPSLAM/feature/line_descriptor/binary_descriptor_matcher.cpp:197-255; Mihasher(256, 32) :597-818).
// Multi-index hashing returns, among the train descriptors at the minimum Hamming distance, the one it DISCOVERS
// first: search radius s ascending, substring (byte) k ascending, then the order in which the s-bit flip patterns
// are enumerated, then ascending train index inside a bucket.  A descriptor is first discovered at
// (s, k) = lexicographic minimum over k of (popcount(q_k ^ t_k), k).  The kernel brute-forces the distances and
// breaks ties with exactly that key; `rank` = position of an 8-bit flip pattern in the enumeration of its
// popcount class (restated on the host).  Nothing within distance 128 -> (-1, 256): the reference reads
// uninitialised memory there.
// grid = (ceil(nq_cap / 4), B), block = 256: one wave per query line.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lbd_match_1nn(const uint8_t* __restrict__ q, const int32_t* __restrict__ q_counts, int nq_cap,
                                                       const uint8_t* __restrict__ t, const int32_t* __restrict__ t_counts, int nt_cap,
                                                       MihRanks R, int32_t* __restrict__ out_idx, int32_t* __restrict__ out_dist) {
    const int lane = threadIdx.x & 63, b = blockIdx.y, qi = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nq = q_counts ? min(q_counts[b], nq_cap) : nq_cap, nt = t_counts ? min(t_counts[b], nt_cap) : nt_cap;
    if (qi >= nq) return;
    const uint8_t* Q = q + ((size_t)b * nq_cap + qi) * 32;
    const uint8_t* T = t + (size_t)b * nt_cap * 32;
    const uint4 q0 = reinterpret_cast<const uint4*>(Q)[0], q1 = reinterpret_cast<const uint4*>(Q)[1];
    unsigned long long best = ~0ull;
    for (int i = lane; i < nt; i += 64) {
        const uint4 d0 = reinterpret_cast<const uint4*>(T + 32 * (size_t)i)[0], d1 = reinterpret_cast<const uint4*>(T + 32 * (size_t)i)[1];
        const uint32_t x[8] = {q0.x ^ d0.x, q0.y ^ d0.y, q0.z ^ d0.z, q0.w ^ d0.w, q1.x ^ d1.x, q1.y ^ d1.y, q1.z ^ d1.z, q1.w ^ d1.w};
        unsigned dist = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) dist += __popc(x[w]);
        // discovery key: smallest (byte distance, byte index), then the flip pattern's enumeration rank
        unsigned disc = 0xffffffffu;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const unsigned pat = (x[k >> 2] >> (8 * (k & 3))) & 0xffu;
            const unsigned key = ((unsigned)__popc(pat) << 16) | ((unsigned)k << 8) | R.rank[pat];
            disc = min(disc, key);
        }
        const unsigned long long key = ((unsigned long long)dist << 40) | ((unsigned long long)disc << 16) | (unsigned)i;
        best = key < best ? key : best;
    }
    best = wave_min_u64(best);
    if (lane == 0) {
        const size_t o = (size_t)b * nq_cap + qi;
        const unsigned dist = (unsigned)(best >> 40);
        if (best == ~0ull || dist > 128u) { out_idx[o] = -1; out_dist[o] = 256; }
        else { out_idx[o] = (int32_t)(best & 0xffff); out_dist[o] = (int32_t)dist; }
    }
}

void launch_lbd_match_1nn(hipStream_t st, const uint8_t* q, const int32_t* q_counts, int nq_cap, const uint8_t* t, const int32_t* t_counts,
                          int nt_cap, const MihRanks& R, int32_t* out_idx, int32_t* out_dist, int B) {
    hipLaunchKernelGGL(k_lbd_match_1nn, dim3((nq_cap + 3) / 4, B), dim3(256), 0, st, q, q_counts, nq_cap, t, t_counts, nt_cap, R, out_idx, out_dist);
}

}  // namespace plp
