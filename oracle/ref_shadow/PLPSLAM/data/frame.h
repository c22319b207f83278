// TEST INFRASTRUCTURE ONLY (oracle/ref_shadow): shadows src/PLPSLAM/data/frame.h with the public members the reference's
// matcher sources read and write (data/frame.h:262-372).  The two grid getters forward to the REFERENCE'S data/common.cc
// exactly as data/frame.cc:881-893 does.
#ifndef PLPSLAM_DATA_FRAME_H
#define PLPSLAM_DATA_FRAME_H
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
class frame {
public:
    camera::base* camera_ = nullptr;
    unsigned int num_keypts_ = 0;
    std::vector<cv::KeyPoint> keypts_, undist_keypts_;
    eigen_alloc_vector<Vec3_t> bearings_;
    std::vector<float> stereo_x_right_, depths_;
    cv::Mat descriptors_;
    DBoW2::BowVector bow_vec_;
    DBoW2::FeatureVector bow_feat_vec_;
    std::vector<landmark*> landmarks_;
    std::vector<bool> outlier_flags_;
    std::vector<std::vector<std::vector<unsigned int>>> keypt_indices_in_cells_;
    Mat44_t cam_pose_cw_ = Mat44_t::Identity();
    unsigned int num_scale_levels_ = 8;
    float scale_factor_ = 1.2f, log_scale_factor_ = 0.1823216f;
    std::vector<float> scale_factors_, inv_scale_factors_, level_sigma_sq_, inv_level_sigma_sq_;
    // FW: line members
    unsigned int _num_keylines = 0;
    std::vector<cv::line_descriptor::KeyLine> _keylsd;
    std::vector<std::pair<float, float>> _stereo_x_right_cooresponding_to_keylines;
    cv::Mat _lbd_descr;
    std::vector<Line*> _landmarks_line;
    std::vector<bool> _outlier_flags_line;
    unsigned int _num_scale_levels_lsd = 2;
    float _scale_factor_lsd = 2.f, _log_scale_factor_lsd = 0.6931472f;
    std::vector<float> _scale_factors_lsd, _inv_scale_factors_lsd, _level_sigma_sq_lsd, _inv_level_sigma_sq_lsd;

    void assign_grid() { assign_keypoints_to_grid(camera_, undist_keypts_, keypt_indices_in_cells_); }   // data/frame.cc (every constructor)
    std::vector<unsigned int> get_keypoints_in_cell(const float ref_x, const float ref_y, const float margin, const int min_level = -1, const int max_level = -1) const {
        return data::get_keypoints_in_cell(camera_, undist_keypts_, keypt_indices_in_cells_, ref_x, ref_y, margin, min_level, max_level);
    }
    std::vector<unsigned int> get_keylines_in_cell(const float ref_x1, const float ref_y1, const float ref_x2, const float ref_y2, const float margin,
                                                   const int min_level = -1, const int max_level = -1) const {
        return data::get_keylines_in_cell(_keylsd, ref_x1, ref_y1, ref_x2, ref_y2, margin, min_level, max_level);
    }
};
}  // namespace data
}  // namespace PLPSLAM
#endif
