// TEST INFRASTRUCTURE ONLY (oracle/ref_shadow): shadows src/PLPSLAM/data/keyframe.h (see frame.h next to it; the grid
// getters forward as data/keyframe.cc:575-587 does).
#ifndef PLPSLAM_DATA_KEYFRAME_H
#define PLPSLAM_DATA_KEYFRAME_H
#include <set>
#include <utility>
#include <vector>

#include <opencv2/core.hpp>
#include <DBoW2/BowVector.h>
#include <DBoW2/FeatureVector.h>

#include "PLPSLAM/type.h"
#include "PLPSLAM/camera/base.h"
#include "PLPSLAM/data/common.h"
#include "PLPSLAM/data/landmark.h"
#include "PLPSLAM/data/landmark_line.h"

namespace PLPSLAM {
namespace data {
class keyframe {
public:
    unsigned int id_ = 0;
    camera::base* camera_ = nullptr;
    unsigned int num_keypts_ = 0;
    std::vector<cv::KeyPoint> keypts_, undist_keypts_;
    eigen_alloc_vector<Vec3_t> bearings_;
    std::vector<float> stereo_x_right_, depths_;
    cv::Mat descriptors_;
    DBoW2::BowVector bow_vec_;
    DBoW2::FeatureVector bow_feat_vec_;
    std::vector<std::vector<std::vector<unsigned int>>> keypt_indices_in_cells_;
    unsigned int num_scale_levels_ = 8;
    float scale_factor_ = 1.2f, log_scale_factor_ = 0.1823216f;
    std::vector<float> scale_factors_, level_sigma_sq_, inv_level_sigma_sq_;
    unsigned int _num_keylines = 0;
    std::vector<cv::line_descriptor::KeyLine> _keylsd;
    std::vector<std::pair<float, float>> _stereo_x_right_cooresponding_to_keylines;
    cv::Mat _lbd_descr;
    unsigned int _num_scale_levels_lsd = 2;
    float _scale_factor_lsd = 2.f, _log_scale_factor_lsd = 0.6931472f;
    std::vector<float> _scale_factors_lsd, _level_sigma_sq_lsd, _inv_level_sigma_sq_lsd;
    // ---- preset by the driver
    Mat44_t cam_pose_cw_ = Mat44_t::Identity();
    std::vector<landmark*> landmarks_;
    std::vector<Line*> landmarks_line_;

    void assign_grid() { assign_keypoints_to_grid(camera_, undist_keypts_, keypt_indices_in_cells_); }
    Mat33_t get_rotation() const { return cam_pose_cw_.block<3, 3>(0, 0); }
    Vec3_t get_translation() const { return cam_pose_cw_.block<3, 1>(0, 3); }
    Vec3_t get_cam_center() const { return -get_rotation().transpose() * get_translation(); }
    std::vector<landmark*> get_landmarks() const { return landmarks_; }
    std::vector<Line*> get_landmarks_line() const { return landmarks_line_; }
    std::set<landmark*> get_valid_landmarks() const {
        std::set<landmark*> v;
        for (auto* lm : landmarks_) if (lm && !lm->will_be_erased()) v.insert(lm);
        return v;
    }
    landmark* get_landmark(const unsigned int idx) const { return landmarks_.at(idx); }
    Line* get_landmark_line(const unsigned int idx) const { return landmarks_line_.at(idx); }
    void add_landmark(landmark* lm, const unsigned int idx) { landmarks_.at(idx) = lm; }
    void add_landmark_line(Line* line, const unsigned int idx) { landmarks_line_.at(idx) = line; }
    std::vector<unsigned int> get_keypoints_in_cell(const float ref_x, const float ref_y, const float margin) const {
        return data::get_keypoints_in_cell(camera_, undist_keypts_, keypt_indices_in_cells_, ref_x, ref_y, margin);
    }
    std::vector<unsigned int> get_keylines_in_cell(const float ref_x1, const float ref_y1, const float ref_x2, const float ref_y2, const float margin,
                                                   const int min_level = -1, const int max_level = -1) const {
        return data::get_keylines_in_cell(_keylsd, ref_x1, ref_y1, ref_x2, ref_y2, margin, min_level, max_level);
    }
};
}  // namespace data
}  // namespace PLPSLAM
#endif
