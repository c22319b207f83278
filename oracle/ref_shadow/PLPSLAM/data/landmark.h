// TEST INFRASTRUCTURE ONLY (oracle/ref_shadow): shadows src/PLPSLAM/data/landmark.h with the members the reference's
// matcher sources touch.  The host-side geometry (predict_scale_level, distances, normals) is preset per landmark by the
// driver: those are INPUTS of the searches that oracle/_ref pins, as they are inputs of the array-form oracle.
#ifndef PLPSLAM_DATA_LANDMARK_H
#define PLPSLAM_DATA_LANDMARK_H
#include <limits>
#include <utility>
#include <vector>

#include <opencv2/core/core.hpp>

#include "PLPSLAM/type.h"

namespace PLPSLAM {
namespace data {
class frame;
class keyframe;
class landmark {
public:
    // tracking_module::search_local_landmarks state (data/landmark.h:150-153)
    Vec2_t reproj_in_tracking_;
    float x_right_in_tracking_ = -1.f;
    bool is_observable_in_tracking_ = false;
    int scale_level_in_tracking_ = 0;
    // ---- preset by the driver
    int id_ = -1;
    Vec3_t pos_w_;
    cv::Mat desc_;
    bool erased_ = false;
    unsigned int num_obs_ = 0;
    float min_dist_ = 0.f, max_dist_ = std::numeric_limits<float>::max();
    Vec3_t mean_normal_;
    unsigned int pred_level_ = 0;
    const keyframe* observed_in_ = nullptr;   // the one key frame in which this landmark counts as observed
    int index_in_observed_ = -1;
    // ---- recorded for the driver
    landmark* replaced_by_ = nullptr;
    std::vector<std::pair<keyframe*, unsigned int>> added_observations_;

    Vec3_t get_pos_in_world() const { return pos_w_; }
    Vec3_t get_obs_mean_normal() const { return mean_normal_; }
    cv::Mat get_descriptor() const { return desc_.clone(); }
    unsigned int num_observations() const { return num_obs_; }
    bool has_observation() const { return 0 < num_obs_; }
    bool will_be_erased() { return erased_; }
    bool is_observed_in_keyframe(keyframe* keyfrm) const { return observed_in_ != nullptr && observed_in_ == keyfrm; }
    int get_index_in_keyframe(keyframe* keyfrm) const { return (observed_in_ != nullptr && observed_in_ == keyfrm) ? index_in_observed_ : -1; }
    float get_min_valid_distance() const { return min_dist_; }
    float get_max_valid_distance() const { return max_dist_; }
    unsigned int predict_scale_level(const float, const frame*) const { return pred_level_; }
    unsigned int predict_scale_level(const float, const keyframe*) const { return pred_level_; }
    void add_observation(keyframe* keyfrm, unsigned int idx) { added_observations_.emplace_back(keyfrm, idx); ++num_obs_; }
    void replace(landmark* lm) { replaced_by_ = lm; erased_ = true; }
};
}  // namespace data
}  // namespace PLPSLAM
#endif
