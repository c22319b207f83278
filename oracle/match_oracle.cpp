// TEST INFRASTRUCTURE ONLY (oracle).  Not part of the product path.
//
// CPU restatement, in array form, of the reference's Hamming matchers
//   src/PLPSLAM/match/base.h:43-92                 compute_descriptor_distance_32/64
//   src/PLPSLAM/match/angle_checker.h:38-176       30-bin delta-angle histogram, top-3 bins valid
//   src/PLPSLAM/data/common.h:104-109, common.cc:205-313   64x48 key-point grid and its window query
//   src/PLPSLAM/match/projection.cc:37-121         match_frame_and_landmarks
//   src/PLPSLAM/match/projection.cc:214-358        match_current_and_last_frames
//   src/PLPSLAM/match/robust.cc:257-385            brute_force_match
// "Array form": the reference walks data::frame / data::landmark objects; here every quantity those
// loops read is a plain array prepared by the caller (reprojections, predicted levels, descriptors,
// occupancy flags), and every quantity they write comes back as an array.  Iteration order, skip
// rules, thresholds and tie behaviour follow the cited lines.
//
// Pinning: test/PLPSLAM/match/base.cc (3 Hamming vectors), test/PLPSLAM/match/angle_checker.cc and
// test/PLPSLAM/data/common_get_cell_indices.cc are replayed in tests/test_oracle_match.py; the
// matcher functions themselves have no tests in the reference -> "parity unpinned" for them.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <unordered_set>
#include <vector>

#include "cv_restated.hpp"

namespace oracle {

constexpr unsigned HAMMING_DIST_THR_LOW = 50, HAMMING_DIST_THR_HIGH = 100, MAX_HAMMING_DIST = 256;

inline unsigned hamming32(const uint8_t* a, const uint8_t* b) {  // base.h:43-68 (SWAR on 8 x u32)
    uint32_t pa[8], pb[8];
    std::memcpy(pa, a, 32); std::memcpy(pb, b, 32);
    unsigned dist = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t v = pa[i] ^ pb[i];
        v -= ((v >> 1) & 0x55555555U);
        v = (v & 0x33333333U) + ((v >> 2) & 0x33333333U);
        dist += (((v + (v >> 4)) & 0x0F0F0F0FU) * 0x01010101U) >> 24;
    }
    return dist;
}
inline unsigned hamming64(const uint8_t* a, const uint8_t* b) {  // base.h:70-92
    uint64_t pa[4], pb[4];
    std::memcpy(pa, a, 32); std::memcpy(pb, b, 32);
    unsigned dist = 0;
    for (int i = 0; i < 4; ++i) {
        uint64_t v = pa[i] ^ pb[i];
        v -= (v >> 1) & 0x5555555555555555UL;
        v = (v & 0x3333333333333333UL) + ((v >> 2) & 0x3333333333333333UL);
        dist += (unsigned)((((v + (v >> 4)) & 0x0F0F0F0F0F0F0F0FUL) * 0x0101010101010101UL) >> 56);
    }
    return dist;
}

// angle_checker<int> (angle_checker.h): bins = cvRound(delta/30) (only 0..12 reachable), invalid =
// matches outside the 3 fullest bins.  The reference orders bins with std::sort and a size-only
// comparator (angle_checker.h:165-176), so which of several EQUALLY full bins makes the top 3 is
// unspecified there.  Deliberate definition (oracle and HIP path alike): equally full bins keep
// ascending bin order (a stable sort).
// set by AngleChecker::collect: the thr-th and (thr+1)-th fullest bins held equally many (> 0) matches, i.e. the order std::sort
// gives equal elements (angle_checker.h:165-176) decides which of them is kept (D3); read through oracle_angle_checker_last_tie()
// by tests that want to know how often that happens
static int g_angle_tie = 0;
struct AngleChecker {
    std::vector<std::vector<int>> hist;
    unsigned len, thr;
    float inv_len;
    explicit AngleChecker(unsigned l = 30, unsigned t = 3) : hist(l), len(l), thr(t), inv_len(1.0f / l) {}
    void append(float delta, int match) {
        if (delta < 0.0) delta += 360.0;
        if (360.0 <= delta) delta -= 360.0;
        const unsigned bin = (unsigned)cv_round(delta * inv_len);
        hist.at(bin).push_back(match);
    }
    std::vector<unsigned> order() const {
        std::vector<unsigned> idx(hist.size());
        std::iota(idx.begin(), idx.end(), 0);
        // std::sort, as the reference (angle_checker.h:165-176): among equally full bins the library's algorithm decides.  The HIP
        // path reproduces libstdc++'s std::sort (csrc/libstdcxx_sort.hpp), so both equal a reference built with GCC's library.
        std::sort(idx.begin(), idx.end(), [&](unsigned a, unsigned b) { return hist.at(a).size() > hist.at(b).size(); });
        return idx;
    }
    std::vector<int> collect(bool valid) const {
        std::vector<int> out;
        const auto bins = order();
        g_angle_tie = (thr < len && thr > 0 && !hist[bins[thr]].empty() && hist[bins[thr - 1]].size() == hist[bins[thr]].size()) ? 1 : 0;
        for (unsigned bin = 0; bin < len; ++bin) {
            const bool is_valid = std::any_of(bins.begin(), bins.begin() + thr, [bin](unsigned i) { return bin == i; });
            if (is_valid == valid) out.insert(out.end(), hist[bin].begin(), hist[bin].end());
        }
        return out;
    }
};

// ---- grid (camera/base.h:91 64 x 48 cells over the undistorted image bounds)
struct Grid {
    float min_x, min_y;            // image_bounds members are float (camera/base.h:78-81)
    double inv_cell_w, inv_cell_h;  // inv_cell_width_/height_ are double (camera/base.h:158-160)
    int cols, rows;
};
inline bool cell_of(const Grid& g, float x, float y, int& cx, int& cy) {  // common.h:104-109
    cx = cv_floor((x - g.min_x) * g.inv_cell_w);
    cy = cv_floor((y - g.min_y) * g.inv_cell_h);
    return 0 <= cx && cx < g.cols && 0 <= cy && cy < g.rows;
}
struct Features {   // what the matchers read from data::frame
    int n;
    const KeyPoint* kps;        // undist_keypts_ (x, y, octave, angle)
    const uint8_t* desc;        // n x 32
    const float* x_right;       // stereo_x_right_ (<= 0: none)
    std::vector<std::vector<std::vector<unsigned>>> cells;   // [col][row] -> indices (ascending)
};
inline void assign_to_grid(const Grid& g, Features& f) {  // common.cc:205-231
    f.cells.assign(g.cols, std::vector<std::vector<unsigned>>(g.rows));
    for (int i = 0; i < f.n; ++i) {
        int cx, cy;
        if (cell_of(g, f.kps[i].x, f.kps[i].y, cx, cy)) f.cells[cx][cy].push_back((unsigned)i);
    }
}
inline std::vector<unsigned> keypoints_in_cell(const Grid& g, const Features& f, float ref_x, float ref_y, float margin,
                                               int min_level, int max_level) {  // common.cc:241-313
    std::vector<unsigned> idx;
    const int min_cx = std::max(0, cv_floor((ref_x - g.min_x - margin) * g.inv_cell_w));
    if (g.cols <= min_cx) return idx;
    const int max_cx = std::min(g.cols - 1, cv_ceil((ref_x - g.min_x + margin) * g.inv_cell_w));
    if (max_cx < 0) return idx;
    const int min_cy = std::max(0, cv_floor((ref_y - g.min_y - margin) * g.inv_cell_h));
    if (g.rows <= min_cy) return idx;
    const int max_cy = std::min(g.rows - 1, cv_ceil((ref_y - g.min_y + margin) * g.inv_cell_h));
    if (max_cy < 0) return idx;
    const bool check_level = (0 < min_level) || (0 <= max_level);
    for (int cx = min_cx; cx <= max_cx; ++cx)
        for (int cy = min_cy; cy <= max_cy; ++cy)
            for (unsigned i : f.cells[cx][cy]) {
                const KeyPoint& k = f.kps[i];
                if (check_level) {
                    if (k.octave < min_level) continue;
                    if (0 <= max_level && max_level < k.octave) continue;
                }
                const float dx = k.x - ref_x, dy = k.y - ref_y;
                if (std::abs(dx) < margin && std::abs(dy) < margin) idx.push_back(i);
            }
    return idx;
}

}  // namespace oracle

using namespace oracle;

extern "C" {

int oracle_angle_checker_last_tie() { return g_angle_tie; }
// index_sort_by_size of the reference (std::sort of the bin indices by bin size, descending) for given bin sizes
void oracle_index_sort_by_size(const int* sizes, int n, unsigned* idx) {
    std::vector<unsigned> v(n);
    std::iota(v.begin(), v.end(), 0u);
    std::sort(v.begin(), v.end(), [&](unsigned a, unsigned b) { return sizes[a] > sizes[b]; });
    for (int i = 0; i < n; ++i) idx[i] = v[i];
}
// The real thing the exact seed order is measured against (csrc/seed_sort_model.hpp): this machine's std::sort on 32-bit entries compared by
// bits 20..29, larger first -- the comparator of lsd.cpp's ordered_points (bin only), with the pixel index as payload -- and, to check the
// intermediate state and forced recursion budgets, libstdc++'s own std::__introsort_loop.
void oracle_std_sort_entries(unsigned* e, long n) {
    std::sort(e, e + n, [](unsigned a, unsigned b) { return (a >> 20) > (b >> 20); });
}
int oracle_std_introsort_loop_entries(unsigned* e, long n, int depth_limit) {
#if defined(__GLIBCXX__)
    auto cmp = [](unsigned a, unsigned b) { return (a >> 20) > (b >> 20); };
    if (depth_limit < 0) depth_limit = 2 * std::__lg(n);
    if (n > 0) std::__introsort_loop(e, e + n, (long)depth_limit, __gnu_cxx::__ops::__iter_comp_iter(cmp));
    return 1;
#else
    (void)e; (void)n; (void)depth_limit;
    return 0;      // another C++ library: its std::sort is another algorithm, nothing to compare the restatement with
#endif
}
unsigned oracle_hamming32(const uint8_t* a, const uint8_t* b) { return hamming32(a, b); }
unsigned oracle_hamming64(const uint8_t* a, const uint8_t* b) { return hamming64(a, b); }

// angle_checker: deltas[n] with payload = index; returns the invalid (valid=0) or valid (valid=1) payloads
int oracle_angle_checker(const float* deltas, int n, int hist_len, int n_bins_thr, int valid, int* out) {
    AngleChecker ac((unsigned)hist_len, (unsigned)n_bins_thr);
    for (int i = 0; i < n; ++i) ac.append(deltas[i], i);
    auto v = ac.collect(valid != 0);
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    return (int)v.size();
}

int oracle_get_cell_indices(float min_x, float min_y, double inv_w, double inv_h, int cols, int rows, float x, float y, int* cx, int* cy) {
    Grid g{min_x, min_y, inv_w, inv_h, cols, rows};
    return cell_of(g, x, y, *cx, *cy) ? 1 : 0;
}

int oracle_keypoints_in_cell(const double* grid6, const KeyPoint* kps, int n, float ref_x, float ref_y, float margin, int min_level,
                             int max_level, unsigned* out) {
    Grid g{(float)grid6[0], (float)grid6[1], grid6[2], grid6[3], (int)grid6[4], (int)grid6[5]};
    Features f{n, kps, nullptr, nullptr, {}};
    assign_to_grid(g, f);
    auto v = keypoints_in_cell(g, f, ref_x, ref_y, margin, min_level, max_level);
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    return (int)v.size();
}

// projection::match_frame_and_landmarks (projection.cc:37-121), array form.
//   frame:  kps/desc/x_right[n], occupied[n] (landmarks_[idx] && has_observation()), scale_factors[]
//   landmarks (in local_landmarks order): valid[m] (is_observable_in_tracking_ && !will_be_erased()),
//            reproj[m][2], x_right[m], level[m] (scale_level_in_tracking_), desc[m][32],
//            has_obs[m] (has_observation() of the landmark: whether later landmarks skip its key point)
//   out: kp_landmark[n] = index of the landmark written into frm.landmarks_[idx] (or -1: untouched)
unsigned oracle_match_frame_and_landmarks(const double* grid6, const KeyPoint* kps, const uint8_t* desc, const float* x_right,
                                          const uint8_t* occupied, int n, const float* scale_factors, const uint8_t* lm_valid,
                                          const float* lm_reproj, const float* lm_x_right, const int* lm_level,
                                          const uint8_t* lm_desc, const uint8_t* lm_has_obs, int m, float margin, float lowe_ratio,
                                          int* kp_landmark) {
    Grid g{(float)grid6[0], (float)grid6[1], grid6[2], grid6[3], (int)grid6[4], (int)grid6[5]};
    Features f{n, kps, desc, x_right, {}};
    assign_to_grid(g, f);
    std::vector<uint8_t> occ(occupied, occupied + n);
    for (int i = 0; i < n; ++i) kp_landmark[i] = -1;
    unsigned num_matches = 0;
    for (int l = 0; l < m; ++l) {
        if (!lm_valid[l]) continue;
        const int lvl = lm_level[l];
        const auto cand = keypoints_in_cell(g, f, lm_reproj[2 * l], lm_reproj[2 * l + 1], margin * scale_factors[lvl], lvl - 1, lvl);
        if (cand.empty()) continue;
        unsigned best = MAX_HAMMING_DIST, second = MAX_HAMMING_DIST;
        int best_level = -1, second_level = -1, best_idx = -1;
        for (unsigned idx : cand) {
            if (occ[idx]) continue;
            if (0 < x_right[idx]) {
                const auto err = std::abs(lm_x_right[l] - x_right[idx]);
                if (margin * scale_factors[lvl] < err) continue;
            }
            const unsigned d = hamming32(lm_desc + 32 * (size_t)l, desc + 32 * (size_t)idx);
            if (d < best) { second = best; best = d; second_level = best_level; best_level = kps[idx].octave; best_idx = (int)idx; }
            else if (d < second) { second_level = kps[idx].octave; second = d; }
        }
        if (best <= HAMMING_DIST_THR_HIGH) {
            if (best_level == second_level && best > lowe_ratio * second) continue;
            kp_landmark[best_idx] = l;
            occ[best_idx] = lm_has_obs[l];
            ++num_matches;
        }
    }
    return num_matches;
}

// projection::match_current_and_last_frames (projection.cc:214-358), array form.
//   last frame entries (in idx_last order): valid[m] (landmark && !outlier && reprojects in image),
//       reproj[m][2], x_right[m], octave[m] (last_frm.keypts_[idx].octave), angle[m] (undist angle), desc (landmark descriptor)
//   direction: 0 neither (level-1..level+1), 1 forward (level..num_levels-1), 2 backward (0..level)
//   out: kp_last[n] = idx_last matched to current key point (or -1)
unsigned oracle_match_current_and_last(const double* grid6, const KeyPoint* kps, const uint8_t* desc, const float* x_right,
                                       const uint8_t* occupied, int n, const float* scale_factors, int num_levels,
                                       const uint8_t* valid, const float* reproj, const float* lx_right, const int* loctave,
                                       const float* langle, const uint8_t* ldesc, const uint8_t* l_has_obs, int m, float margin,
                                       int direction, int check_orientation, int* kp_last) {
    Grid g{(float)grid6[0], (float)grid6[1], grid6[2], grid6[3], (int)grid6[4], (int)grid6[5]};
    Features f{n, kps, desc, x_right, {}};
    assign_to_grid(g, f);
    std::vector<uint8_t> occ(occupied, occupied + n);
    for (int i = 0; i < n; ++i) kp_last[i] = -1;
    unsigned num_matches = 0;
    AngleChecker ac;
    for (int l = 0; l < m; ++l) {
        if (!valid[l]) continue;
        const int lvl = loctave[l];
        const float mg = margin * scale_factors[lvl];
        std::vector<unsigned> cand;
        if (direction == 1) cand = keypoints_in_cell(g, f, reproj[2 * l], reproj[2 * l + 1], mg, lvl, num_levels - 1);
        else if (direction == 2) cand = keypoints_in_cell(g, f, reproj[2 * l], reproj[2 * l + 1], mg, 0, lvl);
        else cand = keypoints_in_cell(g, f, reproj[2 * l], reproj[2 * l + 1], mg, lvl - 1, lvl + 1);
        if (cand.empty()) continue;
        unsigned best = MAX_HAMMING_DIST;
        int best_idx = -1;
        for (unsigned idx : cand) {
            if (occ[idx]) continue;
            if (x_right[idx] > 0) {
                const float err = std::fabs(lx_right[l] - x_right[idx]);
                if (mg < err) continue;
            }
            const unsigned d = hamming32(ldesc + 32 * (size_t)l, desc + 32 * (size_t)idx);
            if (d < best) { best = d; best_idx = (int)idx; }
        }
        if (HAMMING_DIST_THR_HIGH < best) continue;
        kp_last[best_idx] = l;
        occ[best_idx] = l_has_obs[l];
        ++num_matches;
        if (check_orientation) ac.append(langle[l] - kps[best_idx].angle, best_idx);
    }
    if (check_orientation)
        for (int bad : ac.collect(false)) { kp_last[bad] = -1; --num_matches; }
    return num_matches;
}

// robust::brute_force_match (robust.cc:257-385), array form.
//   frame 1 (current): desc1[n1], angle1[n1]; keyframe 2: desc2[n2], angle2[n2], valid2[n2] (landmark && !will_be_erased)
//   out: match_2_in_1[n1] (idx_2 or -1); returns num_matches.  The reference then lists (idx_1, idx_2) in idx_1 order.
unsigned oracle_brute_force_match(const uint8_t* desc1, const float* angle1, int n1, const uint8_t* desc2, const float* angle2,
                                  const uint8_t* valid2, int n2, float lowe_ratio, int check_orientation, int* match_2_in_1) {
    for (int i = 0; i < n1; ++i) match_2_in_1[i] = -1;
    std::unordered_set<int> used;
    unsigned num_matches = 0;
    AngleChecker ac;
    for (int i2 = 0; i2 < n2; ++i2) {
        if (!valid2[i2]) continue;
        unsigned best = MAX_HAMMING_DIST, second = MAX_HAMMING_DIST;
        int best_i1 = -1;
        for (int i1 = 0; i1 < n1; ++i1) {
            if (used.count(i1)) continue;
            const unsigned d = hamming32(desc2 + 32 * (size_t)i2, desc1 + 32 * (size_t)i1);
            if (d < best) { second = best; best = d; best_i1 = i1; }
            else if (d < second) second = d;
        }
        if (HAMMING_DIST_THR_LOW < best) continue;
        if (best_i1 < 0) continue;
        if (lowe_ratio * second < static_cast<float>(best)) continue;
        match_2_in_1[best_i1] = i2;
        used.insert(best_i1);
        if (check_orientation) ac.append(angle1[best_i1] - angle2[i2], best_i1);
        ++num_matches;
    }
    if (check_orientation)
        for (int bad : ac.collect(false)) { match_2_in_1[bad] = -1; --num_matches; }
    return num_matches;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------- line variants
// data::get_keylines_in_cell (data/common.cc:315-363): linear scan, both end points within `margin` of the
// projected line; level filter with the `max_level > 0` quirk (:354).
struct KeyLineRec {   // cv::line_descriptor::KeyLine, 68 bytes
    float angle; int class_id, octave; float pt_x, pt_y, response, size;
    float startPointX, startPointY, endPointX, endPointY, sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY;
    float lineLength; int numOfPixels;
};
static std::vector<unsigned> keylines_in_cell(const KeyLineRec* kl, int n, float x1, float y1, float x2, float y2, float margin,
                                              int min_level, int max_level) {
    std::vector<unsigned> idx;
    const double sp[3] = {x1, y1, 1.0}, ep[3] = {x2, y2, 1.0};
    const double l0 = sp[1] * ep[2] - sp[2] * ep[1], l1 = sp[2] * ep[0] - sp[0] * ep[2], l2 = sp[0] * ep[1] - sp[1] * ep[0];
    const bool check_level = (0 < min_level) || (0 <= max_level);
    for (int i = 0; i < n; ++i) {
        const float dsp = (float)((kl[i].startPointX * l0 + kl[i].startPointY * l1 + l2) / std::sqrt(l0 * l0 + l1 * l1));
        const float dep = (float)((kl[i].endPointX * l0 + kl[i].endPointY * l1 + l2) / std::sqrt(l0 * l0 + l1 * l1));
        if (std::abs(dsp) > margin || std::abs(dep) > margin) continue;
        if (check_level) {
            if (kl[i].octave < min_level) continue;
            if (max_level > 0 && kl[i].octave > max_level) continue;
        }
        idx.push_back((unsigned)i);
    }
    return idx;
}

extern "C" {

int oracle_keylines_in_cell(const KeyLineRec* kl, int n, float x1, float y1, float x2, float y2, float margin, int min_level,
                            int max_level, unsigned* out) {
    auto v = keylines_in_cell(kl, n, x1, y1, x2, y2, margin, min_level, max_level);
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    return (int)v.size();
}

// projection::match_frame_and_landmarks_line (projection.cc:124-212), array form.
//   kp_octave[n]: frm.undist_keypts_.at(idx).octave read with the LINE index (quirk, :187,192)
unsigned oracle_match_frame_and_landmarks_line(const KeyLineRec* kl, const uint8_t* lbd, const int* kp_octave, const uint8_t* occupied, int n,
                                               const float* scale_factors_lsd, const uint8_t* lm_valid, const float* lm_sp,
                                               const float* lm_ep, const int* lm_level, const uint8_t* lm_desc,
                                               const uint8_t* lm_has_obs, int m, float margin, float lowe_ratio, int* line_landmark) {
    std::vector<uint8_t> occ(occupied, occupied + n);
    for (int i = 0; i < n; ++i) line_landmark[i] = -1;
    unsigned num_matches = 0;
    if (m == 0) return 0;
    for (int l = 0; l < m; ++l) {
        if (!lm_valid[l]) continue;
        const int lvl = lm_level[l];
        const auto cand = keylines_in_cell(kl, n, lm_sp[2 * l], lm_sp[2 * l + 1], lm_ep[2 * l], lm_ep[2 * l + 1],
                                           margin * scale_factors_lsd[lvl], lvl - 1, lvl);
        if (cand.empty()) continue;
        unsigned best = MAX_HAMMING_DIST, second = MAX_HAMMING_DIST;
        int best_level = -1, second_level = -1, best_idx = -1;
        for (unsigned idx : cand) {
            if (occ[idx]) continue;
            const unsigned d = hamming32(lm_desc + 32 * (size_t)l, lbd + 32 * (size_t)idx);
            if (d < best) { second = best; best = d; second_level = best_level; best_level = kp_octave[idx]; best_idx = (int)idx; }
            else if (d < second) { second_level = kp_octave[idx]; second = d; }
        }
        if (best <= HAMMING_DIST_THR_HIGH) {
            if (best_level == second_level && best > lowe_ratio * second) continue;
            line_landmark[best_idx] = l;
            occ[best_idx] = lm_has_obs[l];
            ++num_matches;
        }
    }
    return num_matches;
}

// projection::match_current_and_last_frames_line (projection.cc:361-527), array form.
//   valid[m]: landmark present && !outlier && the (partial-occlusion) visibility test passed
//   xr_pair[n][2]: _stereo_x_right_cooresponding_to_keylines; is_rgbd: camera setup RGBD
unsigned oracle_match_current_and_last_line(const KeyLineRec* kl, const uint8_t* lbd, const float* xr_pair, const uint8_t* occupied, int n,
                                            const float* scale_factors_lsd, int num_levels_lsd, const uint8_t* valid,
                                            const float* sp, const float* ep, const float* lxr_sp, const float* lxr_ep,
                                            const int* loctave, const uint8_t* ldesc, const uint8_t* l_has_obs, int m, float margin,
                                            int direction, int is_rgbd, int* line_last) {
    std::vector<uint8_t> occ(occupied, occupied + n);
    for (int i = 0; i < n; ++i) line_last[i] = -1;
    unsigned num_matches = 0;
    for (int l = 0; l < m; ++l) {
        if (!valid[l]) continue;
        const int lvl = loctave[l];
        const float mg = margin * scale_factors_lsd[lvl];
        std::vector<unsigned> cand;
        if (direction == 1) cand = keylines_in_cell(kl, n, sp[2 * l], sp[2 * l + 1], ep[2 * l], ep[2 * l + 1], mg, lvl, num_levels_lsd);
        else if (direction == 2) cand = keylines_in_cell(kl, n, sp[2 * l], sp[2 * l + 1], ep[2 * l], ep[2 * l + 1], mg, 0, lvl + 1);
        else cand = keylines_in_cell(kl, n, sp[2 * l], sp[2 * l + 1], ep[2 * l], ep[2 * l + 1], mg, lvl - 1, lvl + 1);
        if (cand.empty()) continue;
        unsigned best = MAX_HAMMING_DIST;
        int best_idx = -1;
        for (unsigned idx : cand) {
            if (occ[idx]) continue;
            if (is_rgbd && xr_pair[2 * idx] > 0 && xr_pair[2 * idx + 1] > 0) {
                const float e_sp = std::fabs(lxr_sp[l] - xr_pair[2 * idx]), e_ep = std::fabs(lxr_ep[l] - xr_pair[2 * idx + 1]);
                if (mg < e_sp || mg < e_ep) continue;
            }
            const unsigned d = hamming32(ldesc + 32 * (size_t)l, lbd + 32 * (size_t)idx);
            if (d < best) { best = d; best_idx = (int)idx; }
        }
        if (HAMMING_DIST_THR_HIGH < best) continue;
        line_last[best_idx] = l;
        occ[best_idx] = l_has_obs[l];
        ++num_matches;
    }
    return num_matches;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------- BoW-guided, fuse, area
extern "C" {

// bow_tree::match_frame_and_keyframe (bow_tree.cc:41-165) and ::match_keyframes (:167-307), array form.
//   The reference walks the two BoW feature vectors (std::map<node id, vector<feature idx>>) in node order; the
//   caller lists the key-frame features in that order (queries) with their node id; frame features of the same node
//   are visited in ascending index (DBoW2 / fbow append feature indices in feature order).
//   q_valid: key-frame feature has a landmark that will not be erased; t_skip: frame-side static skip
//   (match_keyframes: key frame 2 feature without a valid landmark; frame variant: none).
//   out: t_match[n] = query index matched to target t (or -1); returns num_matches.
unsigned oracle_match_bow(const uint8_t* q_desc, const float* q_angle, const int* q_node, const uint8_t* q_valid, int m,
                          const uint8_t* t_desc, const float* t_angle, const int* t_node, const uint8_t* t_skip, int n, float lowe_ratio,
                          int check_orientation, int* t_match) {
    for (int i = 0; i < n; ++i) t_match[i] = -1;
    std::vector<uint8_t> taken(n, 0);
    unsigned num_matches = 0;
    AngleChecker ac;
    for (int q = 0; q < m; ++q) {
        if (!q_valid[q]) continue;
        unsigned best = MAX_HAMMING_DIST, second = MAX_HAMMING_DIST;
        int best_t = -1;
        for (int t = 0; t < n; ++t) {
            if (t_node[t] != q_node[q]) continue;
            if (t_skip && t_skip[t]) continue;
            if (taken[t]) continue;
            const unsigned d = hamming32(q_desc + 32 * (size_t)q, t_desc + 32 * (size_t)t);
            if (d < best) { second = best; best = d; best_t = t; }
            else if (d < second) second = d;
        }
        if (HAMMING_DIST_THR_LOW < best) continue;
        if (lowe_ratio * second < static_cast<float>(best)) continue;
        t_match[best_t] = q;
        taken[best_t] = 1;
        if (check_orientation) ac.append(q_angle[q] - t_angle[best_t], best_t);
        ++num_matches;
    }
    if (check_orientation)
        for (int bad : ac.collect(false)) { t_match[bad] = -1; --num_matches; }
    return num_matches;
}

// fuse::replace_duplication (fuse.cc:169-331), the search part in array form: per landmark (that passed the
// host-side visibility tests) the key point with the smallest Hamming distance inside the window and the
// chi-square gates; the map mutation (:300-324) stays on the host and runs on best_idx in landmark order.
//   reproj_d[m][2]: reprojection as f64 (Vec2_t), x_right[m], pred_level[m] (unsigned), inv_level_sigma_sq[]
//   out: best_idx[m] (-1: none within HAMMING_DIST_THR_LOW)
void oracle_fuse_search(const double* grid6, const KeyPoint* kps, const uint8_t* desc, const float* x_right, int n,
                        const float* scale_factors, const float* inv_level_sigma_sq, const uint8_t* lm_valid, const double* reproj_d,
                        const float* lm_x_right, const unsigned* pred_level, const uint8_t* lm_desc, int m, float margin, int* best_idx) {
    Grid g{(float)grid6[0], (float)grid6[1], grid6[2], grid6[3], (int)grid6[4], (int)grid6[5]};
    Features f{n, kps, desc, x_right, {}};
    assign_to_grid(g, f);
    for (int l = 0; l < m; ++l) {
        best_idx[l] = -1;
        if (!lm_valid[l]) continue;
        const unsigned pred = pred_level[l];
        const auto cand = keypoints_in_cell(g, f, (float)reproj_d[2 * l], (float)reproj_d[2 * l + 1], margin * scale_factors[pred], -1, -1);
        if (cand.empty()) continue;
        unsigned best = MAX_HAMMING_DIST;
        int bi = -1;
        for (unsigned idx : cand) {
            const KeyPoint& k = kps[idx];
            const unsigned scale_level = static_cast<unsigned>(k.octave);
            if (scale_level < pred - 1 || pred < scale_level) continue;   // unsigned arithmetic, as the reference
            if (x_right[idx] >= 0) {
                const double e_x = reproj_d[2 * l] - k.x, e_y = reproj_d[2 * l + 1] - k.y;
                const float e_xr = lm_x_right[l] - x_right[idx];
                const double err = e_x * e_x + e_y * e_y + e_xr * e_xr;
                constexpr float chi_sq_3D = 7.81473;
                if (chi_sq_3D < err * inv_level_sigma_sq[scale_level]) continue;
            } else {
                const double e_x = reproj_d[2 * l] - k.x, e_y = reproj_d[2 * l + 1] - k.y;
                const double err = e_x * e_x + e_y * e_y;
                constexpr float chi_sq_2D = 5.99146;
                if (chi_sq_2D < err * inv_level_sigma_sq[scale_level]) continue;
            }
            const unsigned d = hamming32(lm_desc + 32 * (size_t)l, desc + 32 * (size_t)idx);
            if (d < best) { best = d; bi = (int)idx; }
        }
        if (HAMMING_DIST_THR_LOW < best) continue;
        best_idx[l] = bi;
    }
}

// area::match_in_consistent_area (area.cc:33-153), array form.  prev_matched_pts is updated in place.
unsigned oracle_match_area(const double* grid6, const KeyPoint* kps1, const uint8_t* desc1, int n1, const KeyPoint* kps2,
                           const uint8_t* desc2, int n2, float* prev_pts, int margin, float lowe_ratio, int check_orientation,
                           int* matched_2_in_1) {
    Grid g{(float)grid6[0], (float)grid6[1], grid6[2], grid6[3], (int)grid6[4], (int)grid6[5]};
    Features f2{n2, kps2, desc2, nullptr, {}};
    assign_to_grid(g, f2);
    unsigned num_matches = 0;
    AngleChecker ac;
    for (int i = 0; i < n1; ++i) matched_2_in_1[i] = -1;
    std::vector<unsigned> matched_dists_2(n2, MAX_HAMMING_DIST);
    std::vector<int> matched_1_in_2(n2, -1);
    for (int i1 = 0; i1 < n1; ++i1) {
        const int lvl = kps1[i1].octave;
        if (0 < lvl) continue;
        const auto cand = keypoints_in_cell(g, f2, prev_pts[2 * i1], prev_pts[2 * i1 + 1], (float)margin, lvl, lvl);
        if (cand.empty()) continue;
        unsigned best = MAX_HAMMING_DIST, second = MAX_HAMMING_DIST;
        int best_i2 = -1;
        for (unsigned i2 : cand) {
            const unsigned d = hamming32(desc1 + 32 * (size_t)i1, desc2 + 32 * (size_t)i2);
            if (matched_dists_2[i2] <= d) continue;
            if (d < best) { second = best; best = d; best_i2 = (int)i2; }
            else if (d < second) second = d;
        }
        if (HAMMING_DIST_THR_LOW < best) continue;
        if (second * lowe_ratio < static_cast<float>(best)) continue;
        const int prev_i1 = matched_1_in_2[best_i2];
        if (0 <= prev_i1) { matched_2_in_1[prev_i1] = -1; --num_matches; }
        matched_2_in_1[i1] = best_i2;
        matched_1_in_2[best_i2] = i1;
        matched_dists_2[best_i2] = best;
        ++num_matches;
        if (check_orientation) ac.append(kps1[i1].angle - kps2[best_i2].angle, i1);
    }
    if (check_orientation)
        for (int bad : ac.collect(false))
            if (0 <= matched_2_in_1[bad]) { matched_2_in_1[bad] = -1; --num_matches; }
    for (int i1 = 0; i1 < n1; ++i1)
        if (0 <= matched_2_in_1[i1]) { prev_pts[2 * i1] = kps2[matched_2_in_1[i1]].x; prev_pts[2 * i1 + 1] = kps2[matched_2_in_1[i1]].y; }
    return num_matches;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------- relocalisation / loop / mapping variants
extern "C" {

// projection::match_frame_and_keyframe (projection.cc:529-645), array form.
//   key-frame landmarks in idx order: valid[m] (present, not erased, not already matched, in image, in the ORB distance
//   range), reproj[m][2] (Vec2_t narrowed to float by get_keypoints_in_cell), pred_level[m], angle[m] = keyfrm undist angle
//   occupied[n]: curr_frm.landmarks_[i] != nullptr.  out: kp_lm[n] = idx of the key-frame landmark written to key point i.
unsigned oracle_match_frame_and_keyframe(const double* grid6, const KeyPoint* kps, const uint8_t* desc, const uint8_t* occupied, int n,
                                         const float* scale_factors, const uint8_t* valid, const float* reproj, const unsigned* pred_level,
                                         const float* langle, const uint8_t* ldesc, int m, float margin, unsigned hamm_dist_thr,
                                         int check_orientation, int* kp_lm) {
    Grid g{(float)grid6[0], (float)grid6[1], grid6[2], grid6[3], (int)grid6[4], (int)grid6[5]};
    Features f{n, kps, desc, nullptr, {}};
    assign_to_grid(g, f);
    std::vector<uint8_t> occ(occupied, occupied + n);
    for (int i = 0; i < n; ++i) kp_lm[i] = -1;
    unsigned num_matches = 0;
    AngleChecker ac;
    for (int l = 0; l < m; ++l) {
        if (!valid[l]) continue;
        const unsigned pred = pred_level[l];
        const auto cand = keypoints_in_cell(g, f, reproj[2 * l], reproj[2 * l + 1], margin * scale_factors[pred], (int)(pred - 1), (int)(pred + 1));
        if (cand.empty()) continue;
        unsigned best = MAX_HAMMING_DIST;
        int best_idx = -1;
        for (unsigned idx : cand) {
            if (occ[idx]) continue;
            const unsigned d = hamming32(ldesc + 32 * (size_t)l, desc + 32 * (size_t)idx);
            if (d < best) { best = d; best_idx = (int)idx; }
        }
        if (hamm_dist_thr < best) continue;
        kp_lm[best_idx] = l;
        occ[best_idx] = 1;
        ++num_matches;
        if (check_orientation) ac.append(langle[l] - kps[best_idx].angle, best_idx);
    }
    if (check_orientation)
        for (int bad : ac.collect(false)) { kp_lm[bad] = -1; --num_matches; }
    return num_matches;
}

// projection::match_frame_and_keyframe_line (projection.cc:648-779), array form.
unsigned oracle_match_frame_and_keyframe_line(const KeyLineRec* kl, const uint8_t* lbd, const uint8_t* occupied, int n,
                                              const float* scale_factors_lsd, const uint8_t* valid, const float* sp, const float* ep,
                                              const unsigned* pred_level, const uint8_t* ldesc, int m, float margin,
                                              unsigned hamm_dist_thr, int* line_lm) {
    std::vector<uint8_t> occ(occupied, occupied + n);
    for (int i = 0; i < n; ++i) line_lm[i] = -1;
    unsigned num_matches = 0;
    for (int l = 0; l < m; ++l) {
        if (!valid[l]) continue;
        const unsigned pred = pred_level[l];
        const auto cand = keylines_in_cell(kl, n, sp[2 * l], sp[2 * l + 1], ep[2 * l], ep[2 * l + 1], margin * scale_factors_lsd[pred],
                                           (int)(pred - 1), (int)(pred + 1));
        if (cand.empty()) continue;
        unsigned best = MAX_HAMMING_DIST;
        int best_idx = -1;
        for (unsigned idx : cand) {
            if (occ[idx]) continue;
            const unsigned d = hamming32(ldesc + 32 * (size_t)l, lbd + 32 * (size_t)idx);
            if (d < best) { best = d; best_idx = (int)idx; }
        }
        if (hamm_dist_thr < best) continue;
        line_lm[best_idx] = l;
        occ[best_idx] = 1;
        ++num_matches;
    }
    return num_matches;
}

// projection::match_by_Sim3_transform (projection.cc:781-892), array form.
//   occupied[n]: matched_lms_in_keyfrm[i] != nullptr; valid[m]: the host-side tests of :797-838.
unsigned oracle_match_by_sim3(const double* grid6, const KeyPoint* kps, const uint8_t* desc, const uint8_t* occupied, int n,
                              const float* scale_factors, const uint8_t* valid, const float* reproj, const unsigned* pred_level,
                              const uint8_t* ldesc, int m, float margin, int* kp_lm) {
    Grid g{(float)grid6[0], (float)grid6[1], grid6[2], grid6[3], (int)grid6[4], (int)grid6[5]};
    Features f{n, kps, desc, nullptr, {}};
    assign_to_grid(g, f);
    std::vector<uint8_t> occ(occupied, occupied + n);
    for (int i = 0; i < n; ++i) kp_lm[i] = -1;
    unsigned num_matches = 0;
    for (int l = 0; l < m; ++l) {
        if (!valid[l]) continue;
        const unsigned pred = pred_level[l];
        const auto cand = keypoints_in_cell(g, f, reproj[2 * l], reproj[2 * l + 1], margin * scale_factors[pred], -1, -1);
        if (cand.empty()) continue;
        unsigned best = MAX_HAMMING_DIST;
        int best_idx = -1;
        for (unsigned idx : cand) {
            if (occ[idx]) continue;
            const unsigned scale_level = static_cast<unsigned>(kps[idx].octave);
            if (scale_level < pred - 1 || pred < scale_level) continue;
            const unsigned d = hamming32(ldesc + 32 * (size_t)l, desc + 32 * (size_t)idx);
            if (d < best) { best = d; best_idx = (int)idx; }
        }
        if (HAMMING_DIST_THR_LOW < best) continue;
        kp_lm[best_idx] = l;
        occ[best_idx] = 1;
        ++num_matches;
    }
    return num_matches;
}

// One direction of projection::match_keyframes_mutually (projection.cc:932-1020 / :1028-1118) and the search of
// fuse::detect_duplication (fuse.cc:40-140): independent best key point per landmark, no blocking, no chi-square gate.
//   signed_level != 0: detect_duplication's `const int pred_scale_level`; thr: 100 (mutual) / 50 (detect_duplication)
void oracle_project_best(const double* grid6, const KeyPoint* kps, const uint8_t* desc, int n, const float* scale_factors,
                         const uint8_t* valid, const double* reproj_d, const unsigned* pred_level, const uint8_t* ldesc, int m,
                         float margin, unsigned thr, int signed_level, int* best_idx_out) {
    Grid g{(float)grid6[0], (float)grid6[1], grid6[2], grid6[3], (int)grid6[4], (int)grid6[5]};
    Features f{n, kps, desc, nullptr, {}};
    assign_to_grid(g, f);
    for (int l = 0; l < m; ++l) {
        best_idx_out[l] = -1;
        if (!valid[l]) continue;
        const unsigned pred = pred_level[l];
        const auto cand = keypoints_in_cell(g, f, (float)reproj_d[2 * l], (float)reproj_d[2 * l + 1], margin * scale_factors[pred], -1, -1);
        if (cand.empty()) continue;
        unsigned best = MAX_HAMMING_DIST;
        int bi = -1;
        for (unsigned idx : cand) {
            if (signed_level) {
                const int scale_level = kps[idx].octave, p = (int)pred;
                if (scale_level < p - 1 || p < scale_level) continue;
            } else {
                const unsigned scale_level = static_cast<unsigned>(kps[idx].octave);
                if (scale_level < pred - 1 || pred < scale_level) continue;
            }
            const unsigned d = hamming32(ldesc + 32 * (size_t)l, desc + 32 * (size_t)idx);
            if (d < best) { best = d; bi = (int)idx; }
        }
        if (thr < best) continue;
        best_idx_out[l] = bi;
    }
}

// the cross check at the end of match_keyframes_mutually (projection.cc:1124-1139): pairs (idx_1, idx_2) that chose each other
unsigned oracle_cross_check(const int* idx2_of_1, int n1, const int* idx1_of_2, int* matched_2_in_1) {
    unsigned num = 0;
    for (int i = 0; i < n1; ++i) {
        matched_2_in_1[i] = -1;
        const int i2 = idx2_of_1[i];
        if (i2 < 0) continue;
        if (idx1_of_2[i2] == i) { matched_2_in_1[i] = i2; ++num; }
    }
    return num;
}

// fuse::replace_duplication_line (fuse.cc:335-505), the search part.
void oracle_fuse_search_line(const KeyLineRec* kl, const uint8_t* lbd, int n, const float* scale_factors_lsd, const float* inv_level_sigma_sq_lsd,
                             const uint8_t* valid, const double* sp_d, const double* ep_d, const unsigned* pred_level, const uint8_t* ldesc,
                             int m, float margin, int* best_idx_out) {
    for (int l = 0; l < m; ++l) {
        best_idx_out[l] = -1;
        if (!valid[l]) continue;
        const unsigned pred = pred_level[l];
        const auto cand = keylines_in_cell(kl, n, (float)sp_d[2 * l], (float)sp_d[2 * l + 1], (float)ep_d[2 * l], (float)ep_d[2 * l + 1],
                                           margin * scale_factors_lsd[pred], -1, -1);
        if (cand.empty()) continue;
        unsigned best = MAX_HAMMING_DIST;
        int bi = -1;
        for (unsigned idx : cand) {
            const KeyLineRec& k = kl[idx];
            const unsigned scale_level = static_cast<unsigned>(k.octave);
            // proj_line = (sp, 1) x (ep, 1) in f64
            const double x1 = sp_d[2 * l], y1 = sp_d[2 * l + 1], x2 = ep_d[2 * l], y2 = ep_d[2 * l + 1];
            const double p0 = y1 * 1.0 - 1.0 * y2, p1 = 1.0 * x2 - x1 * 1.0, p2 = x1 * y2 - y1 * x2;
            const double e_sp = (k.startPointX * p0 + k.startPointY * p1 + p2) / std::sqrt(p0 * p0 + p1 * p1);
            const double e_ep = (k.endPointX * p0 + k.endPointY * p1 + p2) / std::sqrt(p0 * p0 + p1 * p1);
            constexpr float chi_sq_2D = 5.99146;
            if (chi_sq_2D < (e_sp * e_sp + e_ep * e_ep) * inv_level_sigma_sq_lsd[scale_level]) continue;
            const unsigned d = hamming32(ldesc + 32 * (size_t)l, lbd + 32 * (size_t)idx);
            if (d < best) { best = d; bi = (int)idx; }
        }
        if (HAMMING_DIST_THR_LOW < best) continue;
        best_idx_out[l] = bi;
    }
}

// robust::match_for_triangulation (robust.cc:43-216) + check_epipolar_constraint (:387-405), array form.
//   Key frame 1 features listed in BoW node order (q_*), key frame 2 features by index (t_*), node ids as in
//   oracle_match_bow.  q_has_lm / t_has_lm: a landmark is attached (skipped); x_right >= 0 marks stereo key points.
//   E_12 row-major; epipole = bearing of camera centre 1 in key frame 2.  The f64 products are evaluated left to
//   right without contraction (this file is built with -ffp-contract=off).
//   out: match_2_of_q[m] (idx_2 or -1) per listed key frame 1 feature; returns num_matches.
unsigned oracle_match_for_triangulation(const uint8_t* q_desc, const float* q_angle, const int* q_node, const uint8_t* q_has_lm,
                                        const float* q_x_right, const int* q_octave, const double* q_bearing, int m,
                                        const uint8_t* t_desc, const float* t_angle, const int* t_node, const uint8_t* t_has_lm,
                                        const float* t_x_right, const double* t_bearing, int n, const float* scale_factors,
                                        const double* E_12, const double* epipole, int check_orientation, int* match_2_of_q) {
    std::vector<uint8_t> matched2(n, 0);
    for (int q = 0; q < m; ++q) match_2_of_q[q] = -1;
    unsigned num_matches = 0;
    AngleChecker ac;
    for (int q = 0; q < m; ++q) {
        if (q_has_lm[q]) continue;
        const bool stereo1 = 0 <= q_x_right[q];
        const double* b1 = q_bearing + 3 * (size_t)q;
        unsigned best = HAMMING_DIST_THR_LOW;
        int best_i2 = -1;
        for (int t = 0; t < n; ++t) {   // keyfrm_2_indices of the same node, ascending
            if (t_node[t] != q_node[q]) continue;
            if (t_has_lm[t]) continue;
            if (matched2[t]) continue;
            const bool stereo2 = 0 <= t_x_right[t];
            const double* b2 = t_bearing + 3 * (size_t)t;
            const unsigned d = hamming32(q_desc + 32 * (size_t)q, t_desc + 32 * (size_t)t);
            if (HAMMING_DIST_THR_LOW < d || best < d) continue;
            if (!stereo1 && !stereo2) {
                const double cos_dist = epipole[0] * b2[0] + epipole[1] * b2[1] + epipole[2] * b2[2];
                constexpr double cos_dist_thr = 0.99862953475;
                if (cos_dist_thr < cos_dist) continue;
            }
            // check_epipolar_constraint
            const double n0 = E_12[0] * b2[0] + E_12[1] * b2[1] + E_12[2] * b2[2];
            const double n1 = E_12[3] * b2[0] + E_12[4] * b2[1] + E_12[5] * b2[2];
            const double n2 = E_12[6] * b2[0] + E_12[7] * b2[1] + E_12[8] * b2[2];
            const double cos_residual = (n0 * b1[0] + n1 * b1[1] + n2 * b1[2]) / std::sqrt(n0 * n0 + n1 * n1 + n2 * n2);
            const double residual_rad = M_PI / 2.0 - std::abs(std::acos(cos_residual));
            constexpr double residual_rad_thr = 0.2 * M_PI / 180.0;
            if (residual_rad < residual_rad_thr * scale_factors[q_octave[q]]) { best_i2 = t; best = d; }
        }
        if (best_i2 < 0) continue;
        matched2[best_i2] = 1;
        match_2_of_q[q] = best_i2;
        ++num_matches;
        if (check_orientation) ac.append(q_angle[q] - t_angle[best_i2], q);
    }
    if (check_orientation)
        for (int bad : ac.collect(false)) { match_2_of_q[bad] = -1; --num_matches; }
    return num_matches;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------- landmark descriptor maintenance
extern "C" {

// landmark::compute_descriptor (data/landmark.cc:181-245; Line::compute_descriptor data/landmark_line.cc:256-320 is the same
// on LBD rows), array form: descs = the num_descs observation descriptors in observation (std::map) order.  Returns
// best_idx: the descriptor whose MEDIAN Hamming distance to all of them (itself included, rank (unsigned)(0.5 * (n - 1)))
// is smallest, first such index.  n == 0 returns -1 (the reference returns before touching descriptor_).
int oracle_landmark_descriptor(const uint8_t* descs, int num_descs) {
    if (num_descs <= 0) return -1;
    std::vector<std::vector<unsigned>> hamm(num_descs, std::vector<unsigned>(num_descs));
    for (int i = 0; i < num_descs; ++i) {
        hamm[i][i] = 0;
        for (int j = i + 1; j < num_descs; ++j) {
            const unsigned d = hamming32(descs + 32 * (size_t)i, descs + 32 * (size_t)j);
            hamm[i][j] = d; hamm[j][i] = d;
        }
    }
    unsigned best_median = MAX_HAMMING_DIST, best_idx = 0;
    for (int idx = 0; idx < num_descs; ++idx) {
        std::vector<unsigned> part(hamm[idx].begin(), hamm[idx].begin() + num_descs);
        std::sort(part.begin(), part.end());
        const unsigned median = part.at(static_cast<unsigned>(0.5 * (num_descs - 1)));
        if (median < best_median) { best_median = median; best_idx = (unsigned)idx; }
    }
    return (int)best_idx;
}

}  // extern "C"
