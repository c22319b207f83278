// TEST INFRASTRUCTURE ONLY (oracle/ref_shadow): shadows src/PLPSLAM/data/landmark_line.h (see landmark.h next to it).
#ifndef PLPSLAM_DATA_LANDMARK_LINE_H
#define PLPSLAM_DATA_LANDMARK_LINE_H
#include <limits>
#include <utility>
#include <vector>

#include <opencv2/features2d.hpp>
#include "PLPSLAM/feature/line_descriptor/line_descriptor_custom.hpp"
#include "PLPSLAM/camera/base.h"
#include "PLPSLAM/camera/perspective.h"
#include <eigen3/Eigen/Dense>
#include <opencv2/calib3d/calib3d.hpp>
#include <opencv2/core/eigen.hpp>
#include <opencv2/imgproc/imgproc.hpp>

#include "PLPSLAM/type.h"

namespace PLPSLAM {
namespace data {
class frame;
class keyframe;
class Line {
public:
    Vec2_t _reproj_in_tracking_sp, _reproj_in_tracking_ep;
    bool _is_observable_in_tracking = false;
    int _scale_level_in_tracking = 0;
    // ---- preset by the driver
    int id_ = -1;
    Vec6_t pos_w_;
    cv::Mat desc_;
    bool erased_ = false;
    unsigned int num_obs_ = 0;
    float min_dist_ = 0.f, max_dist_ = std::numeric_limits<float>::max();
    unsigned int pred_level_ = 0;
    const keyframe* observed_in_ = nullptr;
    int index_in_observed_ = -1;
    // ---- recorded for the driver
    Line* replaced_by_ = nullptr;
    std::vector<std::pair<keyframe*, unsigned int>> added_observations_;

    Vec6_t get_pos_in_world() const { return pos_w_; }
    cv::Mat get_descriptor() const { return desc_.clone(); }
    unsigned int num_observations() const { return num_obs_; }
    bool has_observation() const { return 0 < num_obs_; }
    bool will_be_erased() { return erased_; }
    bool is_observed_in_keyframe(keyframe* keyfrm) const { return observed_in_ != nullptr && observed_in_ == keyfrm; }
    int get_index_in_keyframe(keyframe* keyfrm) const { return (observed_in_ != nullptr && observed_in_ == keyfrm) ? index_in_observed_ : -1; }
    float get_min_valid_distance() const { return min_dist_; }
    float get_max_valid_distance() const { return max_dist_; }
    unsigned int predict_scale_level(const float&, const float&, const unsigned int&) { return pred_level_; }
    void add_observation(keyframe* keyfrm, unsigned int idx) { added_observations_.emplace_back(keyfrm, idx); ++num_obs_; }
    void replace(Line* line) { replaced_by_ = line; erased_ = true; }
};
}  // namespace data
}  // namespace PLPSLAM
#endif
