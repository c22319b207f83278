// Drop-in replacement of the reference header src/PLPSLAM/match/stereo.h (class PLPSLAM::match::stereo, :44-116; built and
// run in the stereo data::frame constructors, data/frame.cc:277-281, 320-324, 370-374): same constructor, same compute().
// The search (row-band candidates, Hamming < 75, 11x11 L1 patch slide on the pyramid level, parabola, median rejection;
// match/stereo.cc:45-301) runs in libplp_front.so on the pyramids the two extractors already hold in HBM: the
// `image_pyramid_` vectors the reference passes in identify the extractors (feature/plp_registry.h).
#ifndef PLPSLAM_MATCH_STEREO_H
#define PLPSLAM_MATCH_STEREO_H

#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include <opencv2/core.hpp>

#include "PLPSLAM/feature/plp_registry.h"
#include "plp_front.h"

namespace PLPSLAM {
namespace match {

class stereo {
public:
    stereo() = delete;

    stereo(const std::vector<cv::Mat>& left_image_pyramid, const std::vector<cv::Mat>& right_image_pyramid,
           const std::vector<cv::KeyPoint>& keypts_left, const std::vector<cv::KeyPoint>& keypts_right, const cv::Mat& descs_left,
           const cv::Mat& descs_right, const std::vector<float>& scale_factors, const std::vector<float>& inv_scale_factors,
           const float focal_x_baseline, const float true_baseline)
        : left_(feature::plp_registry::find(&left_image_pyramid)), right_(feature::plp_registry::find(&right_image_pyramid)),
          keypts_left_(keypts_left), keypts_right_(keypts_right), descs_left_(descs_left), descs_right_(descs_right),
          focal_x_baseline_(focal_x_baseline), true_baseline_(true_baseline) {
        (void)scale_factors; (void)inv_scale_factors;      // the extractors' own tables are the same values (orb_params.cc:58-107)
        if (!left_ || !right_)
            throw std::runtime_error("match::stereo: the image pyramids must be the image_pyramid_ members of two feature::orb_extractor objects");
    }

    virtual ~stereo() = default;

    //! match/stereo.cc:45-150
    void compute(std::vector<float>& stereo_x_right, std::vector<float>& depths) const {
        static_assert(sizeof(cv::KeyPoint) == sizeof(plp_keypoint), "cv::KeyPoint must be the 28-byte POD");
        const int n_l = static_cast<int>(keypts_left_.size()), n_r = static_cast<int>(keypts_right_.size());
        stereo_x_right.assign(static_cast<size_t>(n_l), -1.0f);
        depths.assign(static_cast<size_t>(n_l), -1.0f);
        if (n_l == 0) return;
        const std::vector<unsigned char> dl = rows32(descs_left_, n_l), dr = rows32(descs_right_, n_r);
        const plp_status s = plp_stereo_compute(left_, right_, reinterpret_cast<const plp_keypoint*>(keypts_left_.data()), n_l,
                                                reinterpret_cast<const plp_keypoint*>(keypts_right_.data()), n_r, dl.data(), dr.data(),
                                                focal_x_baseline_, true_baseline_, stereo_x_right.data(), depths.data());
        if (s != PLP_OK) throw std::runtime_error(std::string("plp_front: ") + plp_strerror(s) + ": " + plp_last_error());
    }

private:
    static std::vector<unsigned char> rows32(const cv::Mat& m, int n) {     // descriptors as n contiguous 32-byte rows
        std::vector<unsigned char> d(static_cast<size_t>(n > 0 ? n : 1) * 32);
        for (int i = 0; i < n; ++i) std::memcpy(d.data() + static_cast<size_t>(i) * 32, m.ptr<unsigned char>(i), 32);
        return d;
    }

    plp_orb* left_;
    plp_orb* right_;
    const std::vector<cv::KeyPoint>& keypts_left_;
    const std::vector<cv::KeyPoint>& keypts_right_;
    const cv::Mat& descs_left_;
    const cv::Mat& descs_right_;
    const float focal_x_baseline_;
    const float true_baseline_;
};

}  // namespace match
}  // namespace PLPSLAM

#endif  // PLPSLAM_MATCH_STEREO_H
