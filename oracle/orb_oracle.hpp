// TEST INFRASTRUCTURE ONLY (oracle).  Not part of the product path.
//
// CPU restatement of the reference ORB extractor
//   src/PLPSLAM/feature/orb_extractor.{h,cc}, orb_extractor_node.{h,cc},
//   orb_params.{h,cc}, util/trigonometric.h
// following the scalar default build (USE_SSE_ORB=OFF, USE_OPENMP=OFF, strict
// IEEE f32 without FMA).  OpenCV calls go through cv_restated.hpp.
//
// Pinning status: the reference's own tests hold only property checks for this
// path (test/PLPSLAM/feature/orb_extractor.cc) plus exact scale tables
// (test/PLPSLAM/feature/orb_params.cc:159-211), the per-level quota comment
// (orb_extractor.cc:255-264) and the trig tolerance test; all of those are
// replayed in tests/test_oracle_orb.py.  Where oracle/_ref (the reference's own
// orb_extractor*.cc compiled against oracle/cvshim) is built, the oracle is
// also compared with it keypoint-for-keypoint.  The OpenCV arithmetic itself
// remains "parity unpinned" (see cv_restated.hpp).
//
// One deliberate definition: the reference sorts the forkable-leaf pool by
// (count, node POINTER) (orb_extractor.cc:529), i.e. by heap address.  The
// oracle defines pointer order = creation order (a bump allocator), which is
// what oracle/_ref is also forced to use.
#pragma once
#include <array>
#include <list>
#include <utility>

#include "cv_restated.hpp"

namespace oracle {

struct OrbParams {
    unsigned max_num_keypts = 2000;
    float scale_factor = 1.2f;
    unsigned num_levels = 8;
    unsigned ini_fast_thr = 20;
    unsigned min_fast_thr = 7;
    std::vector<std::array<float, 4>> mask_rects;  // x_min/cols, x_max/cols, y_min/rows, y_max/rows
};

// orb_params.cc:86-128
inline std::vector<float> calc_scale_factors(unsigned n, float sf) {
    std::vector<float> v(n, 1.0f);
    for (unsigned l = 1; l < n; ++l) v[l] = sf * v[l - 1];
    return v;
}
inline std::vector<float> calc_inv_scale_factors(unsigned n, float sf) {
    std::vector<float> v(n, 1.0f);
    for (unsigned l = 1; l < n; ++l) v[l] = (1.0f / sf) * v[l - 1];
    return v;
}
inline std::vector<float> calc_level_sigma_sq(unsigned n, float sf) {
    std::vector<float> v(n, 1.0f);
    float s = 1.0f;
    for (unsigned l = 1; l < n; ++l) { s = sf * s; v[l] = s * s; }
    return v;
}
inline std::vector<float> calc_inv_level_sigma_sq(unsigned n, float sf) {
    std::vector<float> v(n, 1.0f);
    float s = 1.0f;
    for (unsigned l = 1; l < n; ++l) { s = sf * s; v[l] = 1.0f / (s * s); }
    return v;
}

// util/trigonometric.h:31-78 (even polynomial, f32, no FMA)
namespace trig {
constexpr float PI = 3.14159265358979f;
constexpr float PI_2 = PI / 2.0f;
constexpr float TWO_PI = 2.0f * PI;
constexpr float INV_TWO_PI = 1.0f / TWO_PI;
constexpr float THREE_PI_2 = 3.0f * PI_2;
inline float poly(float v) {
    const float v2 = v * v;
    return 0.99940307f + v2 * (-0.49558072f + 0.03679168f * v2);
}
inline float cos(float v) {
    v = v - cv_floor(v * INV_TWO_PI) * TWO_PI;
    v = (0.0f < v) ? v : -v;
    if (v < PI_2) return poly(v);
    if (v < PI) return -poly(PI - v);
    if (v < THREE_PI_2) return -poly(v - PI);
    return poly(TWO_PI - v);
}
inline float sin(float v) { return trig::cos(PI_2 - v); }
}  // namespace trig

extern const int8_t kRbriefPattern[1024];

class OrbOracle {
public:
    explicit OrbOracle(const OrbParams& p) : p_(p) { initialize(); }

    // orb_extractor.cc:73-160
    void extract(const Image& image, const Image* image_mask, std::vector<KeyPoint>& keypts,
                 std::vector<uint8_t>& descriptors);

    void set_max_num_keypoints(unsigned n) { p_.max_num_keypts = n; initialize(); }

    // stage outputs kept for per-stage parity tests
    std::vector<Image> pyramid;                        // image_pyramid_
    std::vector<Image> blurred;                        // per-level blur (empty if level had no keypoints)
    std::vector<std::vector<KeyPoint>> candidates;     // keypts_to_distribute per level (border-relative)
    std::vector<std::vector<KeyPoint>> level_keypts;   // after tree + orientation (level coords)
    std::vector<float> scale_factors, inv_scale_factors, level_sigma_sq, inv_level_sigma_sq;
    std::vector<unsigned> num_keypts_per_level;
    std::vector<int> u_max;
    Image rect_mask;

    static constexpr int kPatch = 31, kHalfPatch = 15, kBorder = 19;

    // exposed for unit tests
    std::vector<KeyPoint> distribute_via_tree(const std::vector<KeyPoint>& kps, int min_x, int max_x,
                                              int min_y, int max_y, unsigned num_keypts) const;
    float ic_angle(const Image& img, float px, float py) const;
    void rbrief(const KeyPoint& kp, const Image& blurred_img, uint8_t* desc) const;

private:
    void initialize();
    void build_pyramid(const Image& image);
    void fast_keypoints(const Image* mask);
    OrbParams p_;
    bool mask_is_initialized_ = false;
};

}  // namespace oracle
